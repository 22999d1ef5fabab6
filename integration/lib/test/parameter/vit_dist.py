# Optional replacement for lib/test/parameter/vit_dist.py of the reference tree.
# Inside the reference tree the paths come from the user's own lib/test/evaluation/local.py, as with the original file.
from vittracker_amd.parameter.vit_dist import parameters as _parameters


def parameters(yaml_name: str):
    from lib.test.evaluation.environment import env_settings
    return _parameters(yaml_name, env=env_settings())
