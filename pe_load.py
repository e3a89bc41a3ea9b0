"""Import helper: registers the directory `phy-engine_amd/` as the package `phy_engine_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "phy_engine_amd" in sys.modules:
        return sys.modules["phy_engine_amd"]
    pkg_dir = os.path.join(ROOT, "phy-engine_amd")
    spec = importlib.util.spec_from_file_location("phy_engine_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["phy_engine_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    """TEST / BASELINE ONLY: the CPU restatement under oracle/."""
    if "pe_oracle" in sys.modules:
        return sys.modules["pe_oracle"]
    spec = importlib.util.spec_from_file_location("pe_oracle", os.path.join(ROOT, "oracle", "pe_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pe_oracle"] = mod
    spec.loader.exec_module(mod)
    return mod
