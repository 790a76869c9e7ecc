"""Import alias: `import amdq3d` loads the package in ./quantization-on-3d-object-detection_amd/."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "quantization-on-3d-object-detection_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
