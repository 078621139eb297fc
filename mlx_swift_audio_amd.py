"""Import shim: the package directory is named `mlx-swift-audio_amd` (not a valid Python identifier),
so this module loads it under the importable name `mlx_swift_audio_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlx-swift-audio_amd")
_spec = importlib.util.spec_from_file_location(
    "mlx_swift_audio_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mlx_swift_audio_amd"] = _mod
_spec.loader.exec_module(_mod)
