"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's pipeline-parallel training step.

Nothing under oracle/ is part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it, and only as the checker / CPU baseline.  The product path (diffusion_pipe_amd/) never imports
this package and has no CPU fallback.

Parity status:
  * Wan DiT block arithmetic: PINNED -- oracle/blocks_ref.py is validated against the reference's own
    models/wan/model.py imported in the build container (oracle/make_golden.py), vectors in tests/golden/.
  * engine semantics (GAS scaling, clip, schedule helpers, balanced partition): restated from DeepSpeed 0.18.4,
    which is not vendored in /root/reference -> PARITY UNPINNED for those pieces (in-tree pieces -- manual
    partition, patched schedule order, clip_grad_norm_, loss functions, bucket arithmetic -- follow the cited lines).
  * SDXL UNet / CLIP, Flux and HunyuanVideo blocks: restated from diffusers / vendor code that is absent from the
    snapshot -> PARITY UNPINNED.
"""
