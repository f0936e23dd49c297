"""
oracle/ -- CPU restatement of the reference's CutMix mean-teacher hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import anything from
here, and there only as the checker (or as the timed CPU baseline) -- never as the thing shipped. The product
package (`cutmix-semisup-seg_amd/`) never imports `oracle` and fails loudly when its HIP library is missing.

Every function cites the reference file:line (relative to the upstream repository root) whose behaviour it
restates. The oracle is pinned by the golden fixtures under `tests/golden/*.npz`, which were produced by
`tests/golden/make_golden.py` *importing the reference's own Python modules* in the build container
(`mask_gen`, `optim_weight_ema`, `evaluation`, `lr_schedules`, `architectures.network_architectures`,
`architectures.deeplab2`). The reference's own test-suite holds no vectors for this path,
so those generated fixtures are the pin; `tests/test_oracle_golden.py` checks every oracle function against
them.

Parts whose arithmetic lives in torchvision 0.5 (DeepLab v3+ backbone/ASPP) have no reference-side vectors:
"parity unpinned" for those.
"""
