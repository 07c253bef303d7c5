from .boxes import GTBoxes  # noqa: F401
from .data_container import DataContainer  # noqa: F401
