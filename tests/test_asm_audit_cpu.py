"""The conv kernel issues asynchronous loads from inline asm (direct-A / mode-3 K loops).  hipcc does not know those loads
are in flight; tools/audit_asm_loads.py proves on the compiled assembly that no compiler-generated instruction touches
a destination register between the load and the hand-counted wait that releases it.

Three checks: the product kernels pass; the audit FLAGS a minimal hazard fixture; and it FLAGS the pre-fix form of the product
K loops that is consistent with the round-1 GPU faults (fault post-mortem): without the `s_waitcnt vmcnt(0)` drain behind
the K loop the last queue refills are still in flight when the epilogue starts, hipcc considers their destination registers dead
and builds the epilogue's store addresses in them -- a late-landing load then overwrites a pointer (memory access fault)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT = os.path.join(ROOT, "tools", "audit_asm_loads.py")
SRC = os.path.join(ROOT, "flickering_adversarial_video_amd", "csrc", "conv_igemm.hip")
pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not available")


def _audit(path):
    return subprocess.run([sys.executable, AUDIT, path], capture_output=True, text=True, timeout=900)


def test_no_compiler_access_to_in_flight_registers():
    r = _audit(SRC)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "kernels with asm loads audited, 0 with violations" in r.stdout and " 0 kernels" not in r.stdout
    # second check of the same run: no compiler-generated vmcnt wait inside a row-ahead tap loop (modes 5 / 6, single and grouped)
    assert "\n0 row-ahead kernels with compiler vmcnt waits in the tap loop" in r.stdout
    assert "conv_igemm_group_kernel" in r.stdout, "the grouped ring kernels issue asm loads too"


def test_producer_consumer_kernel_has_no_register_asm_loads():
    """csrc/conv_pc.hip moves weights AND halo through LDS-DMA (no destination registers: nothing for hipcc's liveness to get wrong); its first
    form kept 33 asm register loads in flight per staging thread.  The audit still runs on it and on its timing-only variants with every
    non-default build (build.ASM_LOAD_SOURCES): a register-destination asm load that comes back must pass it."""
    from concurrent.futures import ThreadPoolExecutor
    pc = os.path.join(ROOT, "flickering_adversarial_video_amd", "csrc", "conv_pc.hip")
    runs = [[], ["-DPC_ABLATE=128"], ["-DPC_STAMP"]]
    with ThreadPoolExecutor(max_workers=3) as ex:
        res = list(ex.map(lambda f: subprocess.run([sys.executable, AUDIT, pc] + f, capture_output=True, text=True, timeout=900), runs))
    for f, r in zip(runs, res):
        assert r.returncode == 0, str(f) + "\n" + r.stdout[-3000:] + r.stderr[-2000:]
        assert "0 kernels with asm loads audited, 0 with violations" in r.stdout, str(f) + r.stdout[-500:]


def test_audit_flags_a_minimal_hazard():
    """tests/fixtures/asm_inflight_fixture.hip: one kernel that touches the destination only behind its wait (must pass), one that
    lets hipcc copy the register while the load is in flight (must be flagged) -- the audit is not vacuous"""
    r = _audit(os.path.join(ROOT, "tests", "fixtures", "asm_inflight_fixture.hip"))
    assert r.returncode == 1, r.stdout + r.stderr
    assert "conv_igemm_kernel_fixture_ok: 1 asm loads, 0 violation(s)" in r.stdout
    assert "conv_igemm_kernel_fixture_bad: 1 asm loads, 2 violation(s)" in r.stdout and "touches in-flight" in r.stdout
    assert "2 kernels with asm loads audited, 1 with violations" in r.stdout


def test_audit_flags_the_pre_fix_k_loops(tmp_path):
    """the product source with the two post-loop drains removed (the state the faulting development builds were in)"""
    s = open(SRC).read()
    drains = ['    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the queue\'s tail before the wave may end\n',
              '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n  } else {']
    assert all(s.count(d) == 1 for d in drains), "conv_igemm.hip changed: update the fixture's patch"
    s = s.replace(drains[0], "").replace(drains[1], "  } else {")
    p = tmp_path / "conv_igemm_nodrain.hip"
    p.write_text(s)
    r = _audit(str(p))
    assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
    # the violations are epilogue index / address arithmetic (v_lshl_add_u64, v_mul_lo_u32, v_mad_u64_u32 ... -- which one depends on
    # hipcc's register allocation of the day) built in queue registers whose tail loads have not landed
    import re
    assert "touches in-flight" in r.stdout
    m = re.search(r"(\d+) kernels with asm loads audited, (\d+) with violations", r.stdout)
    assert m and int(m.group(1)) >= 30 and int(m.group(2)) >= 10, r.stdout[-500:]


def test_audit_flags_a_drained_row_ahead_queue(tmp_path):
    """The audit's second check, on a source that has the hazard: the row-ahead kernels (modes 5 / 6) with their halo staged through REGISTERS
    again (the product stages it by LDS-DMA since round 5) and no wait hipcc can see behind it.  hipcc then carries the halo loads'
    destination registers as "maybe pending" into the tap loop and waits on vmcnt in front of whatever reuses them -- every such wait
    drains the inline-asm weight queue (results correct, the layer 10 % slower: DESIGN_LOG.md, mode 6)."""
    s = open(SRC).read()
    line = "      stage_dma(s);       // (its vmcnt(0) drains the row-ahead weight queue as well: in order, and the halo pieces are the youngest)\n"
    assert s.count(line) == 1, "conv_igemm.hip changed: update the fixture's patch"
    staged = """      {
        const char* src; int ld;
        const bool chvalid = slab_src(s, src, ld);
  #pragma unroll
        for (int n0 = 0; n0 < NPK; n0 += HB) {
          if (n0 * 64 >= p.P) break;
          uint4 v[HB];
  #pragma unroll
          for (int n = 0; n < HB; ++n) v[n] = ldhalo(src, ld, goff[n0 + n], chvalid);
  #pragma unroll
          for (int n = 0; n < HB; ++n)
            if (goff[n0 + n] != -2) *(uint4*)(hdst + (n0 + n) * 1024) = v[n];
        }
      }
"""
    p = tmp_path / "conv_igemm_noscoreboard.hip"
    p.write_text(s.replace(line, staged))
    r = _audit(str(p))
    assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert "compiler wait inside the row-ahead tap loop" in r.stdout
    import re
    m = re.search(r"(\d+) row-ahead kernels with compiler vmcnt waits in the tap loop", r.stdout)
    assert m and int(m.group(1)) >= 1, r.stdout[-500:]
    assert re.search(r"kernels with asm loads audited, 0 with violations", r.stdout), "only the second check may fire"


def test_ablation_variants_pass_the_audit():
    """The round-4 GPU fault: -DCONV_ABLATE variants whose substitute fragments read a weight-queue register before its counted wait.
    The substitutes are zero fragments now; this pins it -- every ablation bit that skips fragment reads (16, 32, both, all) audits clean --
    and nothing is re-run on a GPU to see the abort again.  The four compiles run side by side."""
    from concurrent.futures import ThreadPoolExecutor
    variants = ["-DCONV_ABLATE=16", "-DCONV_ABLATE=32", "-DCONV_ABLATE=48", "-DCONV_ABLATE=63"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda f: subprocess.run([sys.executable, AUDIT, SRC, f], capture_output=True, text=True, timeout=1200), variants))
    for f, r in zip(variants, res):
        assert r.returncode == 0, f + "\n" + r.stdout[-3000:] + r.stderr[-2000:]
        assert "kernels with asm loads audited, 0 with violations" in r.stdout and " 0 kernels" not in r.stdout, f


def test_variant_builds_refuse_a_flagged_source(tmp_path, monkeypatch):
    """build.audit_asm_sources is what FLK_HIPCC_EXTRA builds and tools/build_variant.py call before compiling: it raises on the hazard
    fixture (one kernel lets hipcc copy an in-flight register) and passes a source without asm loads untouched"""
    from flickering_adversarial_video_amd import build as B
    fixture = os.path.join(ROOT, "tests", "fixtures", "asm_inflight_fixture.hip")
    monkeypatch.setattr(B, "CSRC", os.path.dirname(fixture))
    monkeypatch.setattr(B, "ASM_LOAD_SOURCES", ["asm_inflight_fixture.hip"])
    with pytest.raises(RuntimeError, match="audit of asm_inflight_fixture.hip"):
        B.audit_asm_sources(["asm_inflight_fixture.hip"], ["-DSOMETHING=1"])
    B.audit_asm_sources(["pool.hip"], ["-DSOMETHING=1"])      # not an asm-load source: nothing to audit, no exception
