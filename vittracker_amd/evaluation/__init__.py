from .data import Sequence, SequenceList, get_dataset  # noqa: F401
from .tracker import Tracker, trackerlist  # noqa: F401
