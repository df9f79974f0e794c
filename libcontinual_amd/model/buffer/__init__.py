from .linearbuffer import *  # noqa: F401,F403
from .linearherdingbuffer import *  # noqa: F401,F403
from .update import *  # noqa: F401,F403
