"""Importable alias of the package directory `tiered-storage-for-apache-kafka_amd/` (its name is not a Python identifier).

    import tsxform            # == the package in tiered-storage-for-apache-kafka_amd/
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiered-storage-for-apache-kafka_amd")
_spec = importlib.util.spec_from_file_location("tsxform", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tsxform"] = _mod
_spec.loader.exec_module(_mod)
