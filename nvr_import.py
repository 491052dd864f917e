"""Registers the hyphen-named package directory `nano-vllm-rs_amd/` as module `nano_vllm_rs_amd`.

    import nvr_import; nvr = nvr_import.load()
"""
import importlib.util
import os
import sys

_NAME = "nano_vllm_rs_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(root, "nano-vllm-rs_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def load_ctrl():
    """The control-plane module (nano-vllm-rs_amd/ctrl.py: TCP rendezvous of a multi-rank run) — pure Python, loads no library."""
    name = _NAME + "_ctrl"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "nano-vllm-rs_amd", "ctrl.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
