"""oracle/ — TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain torch-CPU, written from scratch) of the reference's algorithm for the hot
path: SemiUHPE's teacher-student training step (hnuzhy/SemiUHPE, `src/agent.py` and what it calls).
Every function cites the reference file:line it follows.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg — as the checker / reported CPU baseline, never as the thing measured or shipped.  The product
package `semiuhpe_amd` never imports it and fails loudly when its HIP library is missing.

Parity status (how each part is pinned):
  * fisher head, rotation adjust, Euler / geodesic metric, CutOut / CutMix, dynamic threshold,
    RepVGG-B1g2, TinyViT-21M, the agent step: PINNED against outputs of the reference itself, run in
    the build container by `tools/make_golden*.py` (import shims only) -> `tests/golden/*.npz`;
    `tests/test_oracle_*.py` check the oracle against those fixtures.
  * EfficientNetV2-S backbone body: **parity unpinned against torchvision** — it lives in torchvision 0.14.1
    (`src/networks.py:301-303`), which is not installed here and not vendored by the reference.
    Restated from the published architecture; pinned as far as the image allows: the
    state-dict key list and shapes (`tests/golden/effnetv2s_state_dict.json`), parameter count (20,177,488
    backbone) and MAC count, and the MBConv block (expand / depthwise / squeeze-excitation / project / skip)
    against an independent implementation on seeded weights -- Hugging Face transformers' EfficientNetBlock
    (`tools/make_golden_mbconv_hf.py` -> `tests/golden/mbconv_hf.npz`).  FusedMBConv, the stage table and
    stochastic depth remain restated only.
  * ResNet-50 body (f3): torchvision absent as well; pinned against transformers' ResNetModel (the same published
    v1.5 network, independent code) on seeded weights renamed key by key (`tools/make_golden_resnet50_hf.py` ->
    `tests/golden/resnet50_hf.npz`): a third-party pin, not the reference's own dependency.
  * pytorch3d 0.7.2 `so3_relative_angle` (cos_bound extrapolation): restated, unpinned.
"""
