#!/usr/bin/env python
"""
Oracle-trained reference runs of the N1 training-parity test (tests/test_gpu_miou_training.py): oracle/step.py =
train_seg_semisup_mask_mt.py:287-467 on PyTorch-CPU fp32, 300 iterations per seed of tests/n1_task.py. Takes ~25 minutes of
CPU per seed, which is why the GPU test does not repeat it: the final teacher mIoU and the supervised-loss log of every seed
are committed as tests/golden/n1_oracle_runs.json (8 CPU threads, torch 2.10: a CPU convolution's reduction order depends on
the thread count, and a 300-step Adam trajectory amplifies that -- the fixture is ONE sample of the reference's own spread).

    python tests/golden/make_n1_oracle_runs.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import torch          # noqa: E402
import n1_task as T   # noqa: E402

if __name__ == '__main__':
    torch.set_num_threads(8)
    seeds = [int(a) for a in sys.argv[1:]] or list(T.SEEDS)
    path = os.path.join(HERE, 'n1_oracle_runs.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    for seed in seeds:
        t0 = time.time()
        miou, log = T.oracle_run(seed, log_every=25)
        out[str(seed)] = dict(miou=miou, sup_loss=[round(v, 6) for v in log], threads=8, iters=T.ITERS, layers=T.LAYERS,
                              seconds=round(time.time() - t0, 1))
        print('seed {}: teacher mIoU {:.4f} ({:.0f} s)'.format(seed, miou, time.time() - t0), flush=True)
        with open(path, 'w') as f:
            json.dump(out, f)
