"""Import alias: the package directory is `directx-renderer-kurth_amd/` (the name the layout contract asks for), which is
not a valid Python identifier; this shim makes it importable as `directx_renderer_kurth_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "directx-renderer-kurth_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
