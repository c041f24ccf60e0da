"""Import shim: makes the in-tree directory ``morbit.jl_amd/`` importable as ``morbit.jl_amd``.

The package directory carries the reference's name (Morbit.jl) with a dot in it, which the
import system cannot address as a plain top-level name; this shim registers it as the
sub-module ``jl_amd`` of ``morbit``.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "morbit.jl_amd")
if "morbit.jl_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        "morbit.jl_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
    )
    jl_amd = importlib.util.module_from_spec(_spec)
    sys.modules["morbit.jl_amd"] = jl_amd
    _spec.loader.exec_module(jl_amd)
else:
    jl_amd = sys.modules["morbit.jl_amd"]
