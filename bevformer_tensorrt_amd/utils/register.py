"""Name -> callable registry with the same surface the reference exposes as
`TRT_FUNCTIONS` (det2trt/models/utils/register.py:9-69, :86): `.get(name)`,
`name in registry`, `.register_module(name=None, force=False, module=None)` usable
directly or as a decorator.  The reference dlopens its TensorRT plugin library at
import time (register.py:72-75); here the HIP library is loaded by `utils.lib`.
"""


class FuncRegistry:
    def __init__(self, name):
        self._name = name
        self._funcs = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._funcs)

    def __len__(self):
        return len(self._funcs)

    def __contains__(self, key):
        return key in self._funcs

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={sorted(self._funcs)})"

    def get(self, key):
        return self._funcs.get(key)

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if not (name is None or isinstance(name, str)):
            raise TypeError(f"name must be None or str, but got {type(name)}")

        def _add(fn):
            key = name or fn.__name__
            if key in self._funcs and not force:
                raise KeyError(f"{key} is already registered in {self._name}")
            self._funcs[key] = fn
            return fn

        return _add(module) if module is not None else _add


TRT_FUNCTIONS = FuncRegistry("tensorrt functions")
