# Drop this file over lib/test/tracker/vit_dist.py of the reference tree:
# the harness imports `lib.test.tracker.vit_dist` and calls get_tracker_class().
from vittracker_amd.tracker.vit_dist import Vit_dist, get_tracker_class  # noqa: F401
