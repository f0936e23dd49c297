#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "tile or epilogue" ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -10
bash tools/gpu_run20.sh
python - <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from cutmix_semisup_seg_amd import ops
DEV='cuda:0'
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator(device=DEV).manual_seed(0)
for name, n, cin, cout, k, dil in (('3x3d2 256->256 n20', 20, 256, 256, 3, 2), ('3x3d2 256->256 n40', 40, 256, 256, 3, 2), ('1x1 1024->256 n20', 20, 1024, 256, 1, 1), ('1x1 256->1024 n20', 20, 256, 1024, 1, 1), ('3x3d4 512->512 n20', 20, 512, 512, 3, 4), ('1x1 2048->512 n40', 40, 2048, 512, 1, 1)):
    pad = dil * (k - 1) // 2
    x = torch.randn(n, 41, 41, cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, cout, cin, generator=g, device=DEV) * 0.05).bfloat16()
    scale, bias = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(n, 41, 41, cout, dtype=torch.bfloat16, device=DEV)
    ts = [timeit(lambda t=t: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, tile=t)) for t in (0, 256, 2256)]
    print(name, ' '.join('%.1f' % t for t in ts), 'us (tile 128 / 256 8-wave / 2256 4-wave 64x128)')
PY
