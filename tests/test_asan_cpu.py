"""AddressSanitizer + LeakSanitizer over the HOST side of libflicker_hip (sanitizers on the CPU build).

build.build_asan_driver() compiles every library source host-only with -fsanitize=address and links it with a heap-backed stand-in
for the HIP runtime (tests/asan/hip_stub.cpp: "device" allocations are malloc blocks, so each upload / memset / read-back the library
issues is an ASan-checked access with the library's own sizes; kernel launches are validated for grid / LDS limits and dropped).
tests/asan/abi_driver.cpp then runs, through the C ABI only: weight packing in every layout, argument validation, and for I3D and
the three VideoResNets plan construction + the complete forward / backward launch sequences + profile read-back + destroy.
No GPU is involved; what is proven is memory safety and leak freedom of the host halves, not kernel arithmetic."""
import os
import struct
import subprocess

import numpy as np
import pytest

from flickering_adversarial_video_amd import build, i3d_spec, videoresnet_spec as vs


def _dump(path, W):
    with open(path, "wb") as f:
        for name, v in W.items():
            a = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<q", a.size))
            f.write(a.tobytes())


@pytest.fixture(scope="module")
def driver():
    return build.build_asan_driver()


def test_host_side_under_asan(driver, tmp_path):
    files = []
    for tag, W in (("i3d", i3d_spec.synthetic_i3d_weights(42)), ("r2plus1d_18", vs.synthetic_weights("r2plus1d_18", 42)),
                   ("r3d_18", vs.synthetic_weights("r3d_18", 42)), ("mc3_18", vs.synthetic_weights("mc3_18", 42))):
        p = str(tmp_path / f"{tag}.bin")
        _dump(p, W)
        files.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", FLK_SINGLE_STREAM="")
    env.pop("FLK_SINGLE_STREAM")
    r = subprocess.run([driver] + files, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + "\n" + r.stderr[-6000:])
    assert "asan driver ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr


def test_asan_catches_a_planted_overflow(driver, tmp_path):
    """the harness is live: a truncated weight file makes the library read past a (too short) host array -- flk_net_set_weight is
    given numel + 1 -- and ASan must report it"""
    src = tmp_path / "probe.cpp"
    src.write_text('#include <vector>\n#include "%s"\nint main(){ flk_net* n=nullptr; flk_net_create(FLK_NET_I3D,FLK_BF16,1,16,224,224,0,&n);'
                   ' std::vector<float> v(8); return flk_net_set_weight(n,"w",v.data(),9); }\n'
                   % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "flicker_hip.h"))
    out_dir = os.path.dirname(driver)
    objs = [os.path.join(out_dir, f) for f in os.listdir(out_dir) if f.endswith(".o") and f != "abi_driver.o"]
    und = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
    defs = sorted({"-Wl,--defsym=" + t + "=0" for line in und.splitlines() for t in line.split() if t.startswith("__hip_fatbin_")})
    exe = str(tmp_path / "probe")
    cc = "/opt/rocm/lib/llvm/bin/clang++"
    subprocess.run([cc, "-fsanitize=address", "-g", "-std=c++17", str(src)] + objs + defs + ["-o", exe, "-ldl", "-lpthread"], check=True,
                   capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr
