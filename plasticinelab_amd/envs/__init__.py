from .env import PlasticineEnv, make          # noqa: F401
from .scenes import ENV_NAMES, load_scene     # noqa: F401
