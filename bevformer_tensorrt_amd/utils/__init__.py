from .register import FuncRegistry, TRT_FUNCTIONS  # noqa: F401
from .lib import load_library, lib_path  # noqa: F401
