"""Build libflicker_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflicker_hip.so")
SOURCES = ["api.cpp", "conv_igemm.hip", "conv_pc.hip", "pool.hip", "head.hip", "attack.hip", "stem_grad.hip", "stem_fwd.hip", "net.cpp", "comm.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# sources that issue asynchronous loads from inline asm (hand-counted s_waitcnt vmcnt): hipcc does not know those loads are in flight
ASM_LOAD_SOURCES = ["conv_igemm.hip", "conv_pc.hip"]


def audit_asm_sources(sources, extra_flags):
    """tools/audit_asm_loads.py on each of `sources` (names under csrc/) that issues inline-asm loads, compiled WITH extra_flags: raises
    when a compiler-generated instruction can touch a register whose asm load is still in flight.  Every non-default build of such a file
    (FLK_HIPCC_EXTRA, tools/build_variant.py) goes through this before anything is linked: the timing-only variants that faulted GPU boxes
    in rounds 1 and 4 were builds the audit would have refused."""
    audit = os.path.join(HERE, "..", "tools", "audit_asm_loads.py")
    for src in sources:
        if src not in ASM_LOAD_SOURCES:
            continue
        r = subprocess.run([sys.executable, audit, os.path.join(CSRC, src)] + list(extra_flags), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"asm-load audit of {src} with flags {list(extra_flags)} FAILED -- not building:\n" + r.stdout[-3000:] + r.stderr[-2000:])


def build(force=False, verbose=True):
    hipcc = _hipcc()
    extra = os.environ.get("FLK_HIPCC_EXTRA", "").split()
    if extra:
        audit_asm_sources(SOURCES, extra)       # a flagged variant never reaches the linker (and so never a GPU)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "flicker_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))


def build_asan_driver(verbose=False):
    """AddressSanitizer build of the library's HOST code (sanitizer stance: sanitizers run on the CPU build only --
    this pool has no GPU ASan): every source compiled with ``--offload-host-only -fsanitize=address`` and linked, together with
    tests/asan/hip_stub.cpp (heap-backed stand-in for the HIP runtime) and tests/asan/abi_driver.cpp, into build/asan/abi_driver.
    Returns the executable's path.  Driven by tests/test_asan_cpu.py."""
    hipcc = _hipcc()
    clangxx = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clangxx):
        clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    root = os.path.dirname(HERE)
    out_dir = os.path.join(root, "build", "asan")
    os.makedirs(out_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(root, "include", "flicker_hip.h")]
    san = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-O1", "-fPIC", "-std=c++17"]
    objs, jobs = [], []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if _stale(o, [s] + headers):
            jobs.append([hipcc, "--offload-arch=gfx950", "--offload-host-only", "-Wno-unused-function", "-x", "hip"] + san + ["-c", s, "-o", o])
    for src in ("hip_stub.cpp", "abi_driver.cpp"):
        s, o = os.path.join(root, "tests", "asan", src), os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if _stale(o, [s] + headers):
            jobs.append([clangxx, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + san + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("asan build failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    exe = os.path.join(out_dir, "abi_driver")
    if jobs or _stale(exe, objs):
        # host-only objects still reference their (absent) device code object: __hip_fatbin_<hash>; the stub's registration ignores it
        und = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
        defs = sorted({"-Wl,--defsym=" + tok + "=0" for line in und.splitlines() for tok in line.split() if tok.startswith("__hip_fatbin_")})
        run([clangxx, "-fsanitize=address"] + objs + defs + ["-o", exe, "-ldl", "-lpthread"])
    return exe
