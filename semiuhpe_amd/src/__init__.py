"""Host-side mirror of the reference's `src/` package (same module paths, function names and argument meaning):
agent.py, networks.py, repvgg.py, tiny_vit.py, augments.py, utils.py, config.py, fisher/fisher_utils.py.
All arithmetic runs in libsuhpe.so (hand-written HIP, gfx950)."""
