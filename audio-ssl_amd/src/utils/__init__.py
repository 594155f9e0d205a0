from src.utils.utils import *  # noqa: F401,F403
