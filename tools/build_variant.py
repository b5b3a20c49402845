#!/usr/bin/env python3
"""Timing-only / experimental variant of libflicker_hip.so: extra compiler flags on ONE source file, the other objects reused.

    tools/build_variant.py NAME [--src conv_igemm.hip] -DCONV_ABLATE=8 ...

writes flickering_adversarial_video_amd/variants/libflk_NAME.so (git-ignored; travels to the GPU box; load it with FLK_LIB_PATH).
Every variant of a source that issues asynchronous loads from inline asm goes through tools/audit_asm_loads.py WITH THE SAME FLAGS
first, and is not linked when the audit finds a compiler-generated access to an in-flight register (the round-1 and round-4 GPU
faults were such variants: DESIGN.md, fault post-mortems).  Build the product library first (python -m ...build)."""
import glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flickering_adversarial_video_amd import build as B  # noqa: E402


def main(argv):
    name, src, flags = argv[0], "conv_igemm.hip", []
    it = iter(argv[1:])
    for a in it:
        if a == "--src":
            src = next(it)
        else:
            flags.append(a)
    pkg = os.path.join(ROOT, "flickering_adversarial_video_amd")
    spath = os.path.join(pkg, "csrc", src)
    B.audit_asm_sources([src], flags)                 # raises on a violation
    out_dir = os.path.join(pkg, "variants")
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.splitext(src)[0]
    obj = os.path.join(out_dir, f"{stem}_{name}.o")
    subprocess.check_call([B._hipcc()] + B.FLAGS + flags + ["-c", spath, "-o", obj])
    others = [o for o in glob.glob(os.path.join(pkg, "csrc", "*.o")) if os.path.basename(o) != stem + ".o"]
    assert len(others) == len(B.SOURCES) - 1, "build the product library first"
    lib = os.path.join(out_dir, f"libflk_{name}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others + ["-ldl"])
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main(sys.argv[1:])
