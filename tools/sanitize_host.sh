#!/bin/bash
# CPU-only sanitizer pass (ASan + UBSan) over the C++ host mirror's file readers, constants and tile plan -- the
# parts of libag2host that run without a GPU.  GPU AddressSanitizer is not available on the pool; the HIP library is
# linked unsanitised and not called.   tools/sanitize_host.sh   (from the repo root, in the build container)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include \
    $R/agile_grasp2_amd/host/ag2_host.cpp $R/tests/cpp/test_host_api.cpp -o $T/drv \
    -L$R/agile_grasp2_amd/csrc -lag2hip -lpthread -Wl,-rpath,$R/agile_grasp2_amd/csrc
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd $R
python - "$T" <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
T = sys.argv[1]
drv = os.path.join(T, "drv")
import test_formats as tf
from agile_grasp2_amd.weights import make_lenet_weights, save_ag2w
w = make_lenet_weights(11)
n = 0
for name, raw in (("new", tf.net_new(w)), ("small", tf.net_new(w, True)), ("v1", tf.net_v1(w))):
    p = os.path.join(T, name + ".caffemodel"); open(p, "wb").write(raw)
    r = subprocess.run([drv, "--caffemodel", p, os.path.join(T, "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    n += 1
    # truncations and bit flips of a valid file: must fail cleanly, never trip a sanitizer
    rng = np.random.default_rng(1)
    for cut in (1, 7, 100, len(raw) // 3, len(raw) - 5):
        q = os.path.join(T, "bad.caffemodel"); open(q, "wb").write(raw[:cut])
        r = subprocess.run([drv, "--caffemodel", q, os.path.join(T, "o.bin")], capture_output=True, text=True)
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        n += 1
    for _ in range(12):
        b = bytearray(raw[:4096]); i = int(rng.integers(0, len(b))); b[i] ^= 1 << int(rng.integers(0, 8))
        q = os.path.join(T, "flip.caffemodel"); open(q, "wb").write(bytes(b) + raw[4096:])
        r = subprocess.run([drv, "--caffemodel", q, os.path.join(T, "o.bin")], capture_output=True, text=True)
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        n += 1
# PCD reader: ascii / binary, with and without normals, then truncated and corrupted headers and bodies
rng = np.random.default_rng(3)
xyz = rng.uniform(-1, 1, size=(300, 3)).astype(np.float32); xyz[7] = np.nan
rgb = rng.integers(0, 2 ** 24, size=300).astype(np.uint32).view(np.float32)
for mode, wn in (("ascii", False), ("binary", False), ("binary", True)):
    p = os.path.join(T, "c.pcd"); tf.write_pcd(p, xyz, rgb, mode, wn)
    raw = open(p, "rb").read()
    variants = [raw] + [raw[:c] for c in (10, 60, 150, len(raw) // 2, len(raw) - 3)]
    variants += [raw.replace(b"POINTS 300", b"POINTS 99999999"), raw.replace(b"WIDTH 300", b"WIDTH -5"),
                 raw.replace(b"SIZE 4", b"SIZE 0", 1), raw.replace(b"FIELDS x y z", b"FIELDS q y z")]
    for k, v in enumerate(variants):
        q = os.path.join(T, "v.pcd"); open(q, "wb").write(v)
        r = subprocess.run([drv, "--pcd", q, os.path.join(T, "o.bin")], capture_output=True, text=True)
        assert (k > 0 or r.returncode == 0) and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (mode, k, r.stderr[-2000:])
        n += 1
r = subprocess.run([drv, "--constants", os.path.join(T, "c.bin")], capture_output=True, text=True)
assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
print("sanitizer pass:", n + 1, "driver runs clean")
PY
rm -rf $T
