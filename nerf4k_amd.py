"""Import alias for the product package.

The package directory is named ``4k-nerf_amd/`` (the framework's name), which is not a legal
Python identifier.  ``import nerf4k_amd`` executes this file, which loads that directory as a
regular package under the name ``nerf4k_amd`` and replaces itself in ``sys.modules``; after
that ``nerf4k_amd.lib.dmpigo`` etc. import normally.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, '4k-nerf_amd')
_spec = importlib.util.spec_from_file_location(
    'nerf4k_amd', os.path.join(_pkg_dir, '__init__.py'), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['nerf4k_amd'] = _mod
_spec.loader.exec_module(_mod)
