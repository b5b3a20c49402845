#!/usr/bin/env python3
"""Static audit of the hand-issued asynchronous loads in conv_igemm.hip / conv_pc.hip (cdna_hip_programming.md, "What hipcc does not do", 1).

The direct-A / mode-3 K loops issue `global_load_dwordx4` from inline asm and retire them with hand-counted
`s_waitcnt vmcnt(N)` statements that name the destination registers ("; release v[a:b] ...").  hipcc treats an asm load's
destination as written when the statement ends, so nothing stops it from copying, spilling or reusing such a register
while the data is still in flight -- which shows up as garbage operands or a memory fault.  This script compiles the file
to assembly and checks, for every kernel, that on no path between an asm load and the wait statement that releases its
destination a compiler-generated instruction reads or writes that register (forward dataflow over the kernel's CFG).

Usage: audit_asm_loads.py [path/to/conv_igemm.hip] [-DFLAG=...]   (exit code 1 on a violation).  Needs hipcc; no GPU."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_asm(src, extra=()):
    hipcc = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc") if c and (os.path.exists(c) or c == "hipcc")), "hipcc")
    out = os.path.join(tempfile.mkdtemp(prefix="flk_audit_"), "k.s")
    inc = [os.path.join(ROOT, "flickering_adversarial_video_amd", "csrc"), os.path.join(ROOT, "include")]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *(f"-I{d}" for d in inc), *extra, "-o", out, src],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def audit_kernel(name, lines):
    """lines: the kernel body.  Forward may-dataflow of the set of registers with an asm load in flight over the kernel's
    control-flow graph (gen: asm load destination; kill: a wait statement that names the register, or any vmcnt(0));
    every compiler-generated instruction is then checked against the set that can reach it.
    Returns (number of asm loads, list of violations)."""
    # ---- instructions: (kind, payload)
    ins, in_asm = [], False
    for raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            ins.append(("label", m.group(1), raw))
            continue
        code = st.split(";")[0].strip()
        if in_asm:
            if "global_load_dwordx4" in st:
                ins.append(("gen", regs_of(st.split(",")[0]), raw))
            elif "s_waitcnt" in st and "vmcnt" in st:
                # vmcnt(0) retires every load; a counted wait retires the registers its "; release ..." comment names
                ins.append(("kill", None if "vmcnt(0)" in st else (regs_of(st.split("release", 1)[1]) if "release" in st else set()), raw))
            continue
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        mb = re.match(r"(s_cbranch\w*|s_branch)\s+(\.LBB\d+_\d+)", code)
        if mb:
            ins.append(("branch", (mb.group(1) == "s_branch", mb.group(2)), raw))
        elif code.startswith("s_endpgm"):
            ins.append(("end", None, raw))
        elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
            ins.append(("kill", None, raw))                  # a compiler drain retires the hand-issued loads as well
        else:
            ins.append(("op", regs_of(code), raw))
    # ---- basic blocks
    starts = {0}
    for i, (k, p, _) in enumerate(ins):
        if k == "label":
            starts.add(i)
        if k in ("branch", "end") and i + 1 < len(ins):
            starts.add(i + 1)
    starts = sorted(starts)
    block_of = {s0: n for n, s0 in enumerate(starts)}
    label_block = {p: block_of[i] for i, (k, p, _) in enumerate(ins) if k == "label"}
    blocks = [(s0, starts[n + 1] if n + 1 < len(starts) else len(ins)) for n, s0 in enumerate(starts)]
    succ = []
    for n, (lo, hi) in enumerate(blocks):
        k, p, _ = ins[hi - 1]
        out = []
        if k == "branch":
            if p[1] in label_block:
                out.append(label_block[p[1]])
            if not p[0] and n + 1 < len(blocks):
                out.append(n + 1)
        elif k != "end" and n + 1 < len(blocks):
            out.append(n + 1)
        succ.append(out)

    def transfer(n, inset, check):
        cur = set(inset)
        for i in range(*blocks[n]):
            k, p, raw = ins[i]
            if k == "gen":
                cur |= p
            elif k == "kill":
                cur = set() if p is None else cur - p
            elif k == "op" and check is not None and cur & p:
                check.append((i, sorted(cur & p), raw.strip()))
        return cur

    inn = [set() for _ in blocks]
    work = [0]
    seen_out = [None] * len(blocks)
    while work:
        n = work.pop()
        out = transfer(n, inn[n], None)
        if seen_out[n] is not None and out <= seen_out[n]:
            continue
        seen_out[n] = out if seen_out[n] is None else seen_out[n] | out
        for m in succ[n]:
            if not out <= inn[m]:
                inn[m] |= out
                work.append(m)
            elif seen_out[m] is None:
                work.append(m)
    violations = []
    for n in range(len(blocks)):
        transfer(n, inn[n], violations)
    return sum(1 for k, _, _ in ins if k == "gen"), violations


ROWAHEAD = re.compile(r"Li[56]EEv(?:6ConvKP|11ConvGroupKP)$")       # conv_igemm_kernel<.., 5 | 6> and the grouped forms


def rowahead_compiler_waits(lines):
    """Second check, row-ahead ring kernels (modes 5 / 6) only: between the first and the last counted `s_waitcnt vmcnt(N > 0) ; release`
    of the tap loop there must be NO compiler-generated vmcnt wait.  hipcc does not see the asm loads, so a wait it inserts for one of
    ITS registers (a halo-load destination it still carries as "maybe pending" behind an exec-masked branch) drains the whole row-ahead
    queue on every step -- correct results, 10 % slower (mode 6).  Returns the offending (line, instruction) pairs."""
    sites, hits, in_asm = [], [], False
    for i, raw in enumerate(lines):
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
        elif st.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            m = re.search(r"s_waitcnt vmcnt\((\d+)\)\s*;\s*release", st)
            if m and int(m.group(1)) > 0:
                sites.append(i)
        else:
            code = st.split(";")[0].strip()
            if code.startswith("s_waitcnt") and "vmcnt" in code:
                hits.append((i, code))
    return [h for h in hits if len(sites) >= 2 and sites[0] < h[0] < sites[-1]]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[1:] if a.startswith("-")]       # e.g. -DCONV_ABLATE=48: audit a timing-only variant BEFORE it runs on a GPU
    src = args[0] if args else os.path.join(ROOT, "flickering_adversarial_video_amd", "csrc", "conv_igemm.hip")
    text = compile_asm(src, extra)
    bad = total = 0
    drained = 0
    # (a kernel's body runs to its .Lfunc_end label: kernels with early exits hold several s_endpgm, each an "end" node of the CFG)
    for m in re.finditer(r"^(\S*(?:conv_igemm_kernel|conv_igemm_group_kernel|conv_pc_kernel|pw_gemm_kernel)\S*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        nloads, viol = audit_kernel(name, body)
        if nloads:
            total += 1
            uniq = sorted({(v[0], tuple(v[1]), v[2]) for v in viol})
            print(f"{name}: {nloads} asm loads, {len(uniq)} violation(s)")
            for i, regs, l in uniq[:10]:
                print(f"    line {i}: touches in-flight v{list(regs)}: {l}")
            bad += bool(uniq)
            if ROWAHEAD.search(name):
                waits = rowahead_compiler_waits(body)
                for i, code in waits[:6]:
                    print(f"    line {i}: compiler wait inside the row-ahead tap loop (drains the queue): {code}")
                drained += bool(waits)
    print(f"{total} kernels with asm loads audited, {bad} with violations")
    print(f"{drained} row-ahead kernels with compiler vmcnt waits in the tap loop")
    return 1 if bad or drained else 0


if __name__ == "__main__":
    sys.exit(main())
