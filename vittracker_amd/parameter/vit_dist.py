"""``parameters(yaml_name)`` for the vit_dist tracker (lib/test/parameter/vit_dist.py:7-30):
merge ``experiments/vit_dist/<yaml_name>.yaml`` into the default config and fill the
TrackerParams the tracker class reads."""
import os

from ..config import cfg, update_config_from_file
from ..evaluation.environment import env_settings
from ..params import TrackerParams


def parameters(yaml_name: str, env=None):
    """`env`: an object with `prj_dir` / `save_dir` (the reference's `env_settings()`, i.e. the user's local.py, when this runs
    inside the reference tree -- integration/lib/test/parameter/vit_dist.py passes it); default: this repo's own settings."""
    params = TrackerParams()
    env = env_settings() if env is None else env
    update_config_from_file(os.path.join(env.prj_dir, "experiments/vit_dist/%s.yaml" % yaml_name))
    params.cfg = cfg
    params.template_factor = cfg.TEST.TEMPLATE_FACTOR
    params.template_size = cfg.TEST.TEMPLATE_SIZE
    params.search_factor = cfg.TEST.SEARCH_FACTOR
    params.search_size = cfg.TEST.SEARCH_SIZE
    params.checkpoint = os.path.join(env.save_dir, "checkpoints/train/vit_dist/%s/OstrackDist_ep%04d.pth.tar"
                                     % (yaml_name, cfg.TEST.EPOCH))
    params.save_all_boxes = False
    return params
