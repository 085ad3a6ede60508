"""Import shim: the product package lives in the directory `mini-sglang_amd/` (the name the
build contract asks for), which is not a valid Python identifier.  `import mini_sglang_amd`
resolves here and re-points the package search path at that directory."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "mini-sglang_amd"
__path__ = [str(_real)]  # submodules (mini_sglang_amd.ops, ...) load from mini-sglang_amd/
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
