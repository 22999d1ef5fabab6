# Optional replacement for lib/test/parameter/vit_dist.py of the reference tree.
from vittracker_amd.parameter.vit_dist import parameters  # noqa: F401
