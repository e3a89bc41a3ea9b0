"""CPU, build container only: the reference's OWN test programs, compiled IN PLACE from /root/reference/test (never copied,
never shipped to the GPU box) against this engine's plug-in API headers (phy-engine_amd/include) and linked with the host
emulation of the kernels (tests/emu, test infrastructure): they must compile unchanged and pass their own assertions.

This is the source-compatibility proof of SURVEY.md 8(b) row 1: model concepts (test/0001.module), netlist operations incl. the
deep copy (0002.net_list), circult::analyze for OP / DC / AC / TR (0003, 0004), every stamper of test/0005.models, the digital /
mixed-signal event loop (0006), the integrators (0008.numerical_methods), Newton + junction limiting (0011) and AC (0012).

The BSIM3v3.2 programs (test/0004.solver/bsim3v32_*.cpp, test/0012.ac/bsim3v32_*.cpp: 106 files) are the host-stamp overlay's hardest
customer (that model stays on the reference's host path): a hook-only model with internal nodes, its own junction
diodes, save_op / load_temperature / check_convergence and complex AC stamps.  The model's header is the REFERENCE's, read where it
lies (`-idirafter /root/reference/include`, used for these programs only); its two relative includes ("../../model_refs/base.h",
"PN_junction.h") are redirected to this repository's headers by a clang VFS overlay written into the build directory -- no reference
source is copied anywhere.  On the GPU the same API runs through tests/cpp (tests/test_gpu_cpp_api.py).
"""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from parity_common import ROOT, make

REF_TESTS = "/root/reference/test"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="the reference tree only exists in the build container")

CXX = "/opt/rocm/lib/llvm/bin/clang++"
EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(ROOT, "tests", "cpp", "_build_ref")

# Expected exit code per program; anything not listed must exit 0.
#   cutthrough.cpp: its second circuit has a resistor of exactly 0 ohm (1/r = inf in the matrix).  The reference's Eigen path
#   returns "success" with NaN node voltages and the program prints them; this engine reports the non-finite system
#   (PE_HIP_ERR_SINGULAR -> analyze() false -> the program's `return -1`).  Documented divergence; the first
#   circuit (r = DBL_MIN) must still print the reference's values.
EXPECTED = {"0005.models/cutthrough.cpp": 255}


def _programs():
    if not os.path.isdir(REF_TESTS):
        return []
    pats = ["0001.module/*.cpp", "0002.net_list/*.cpp", "0003.circuits/*.cpp", "0004.solver/*.cpp", "0005.models/*.cpp", "0006.digital/*.cpp",
            "0008.numerical_methods/*.cpp", "0011.nonlinear/*.cpp", "0012.ac/*.cpp"]
    out = []
    for p in pats:
        out += sorted(os.path.relpath(f, REF_TESTS) for f in glob.glob(os.path.join(REF_TESTS, p)))
    return out


@pytest.fixture(scope="module")
def built():
    make("-C", EMU)
    os.makedirs(OUT, exist_ok=True)
    # the BSIM3 model header includes "../../model_refs/base.h" and "PN_junction.h" relative to ITSELF: map those two paths to this
    # repository's forwarding headers (the plug-in API and the junction model the header is compiled against)
    ref_inc, own_inc = "/root/reference/include/phy_engine/model", f"{ROOT}/phy-engine_amd/include/phy_engine/model"
    vfs = os.path.join(OUT, "bsim3_vfs.yaml")
    with open(vfs, "w") as f:
        f.write("{ 'version': 0, 'case-sensitive': 'true', 'roots': [\n"
                f"  {{ 'type': 'directory', 'name': '{ref_inc}/model_refs', 'contents': [ {{ 'type': 'file', 'name': 'base.h', 'external-contents': '{own_inc}/model_refs/base.h' }} ] }},\n"
                f"  {{ 'type': 'directory', 'name': '{ref_inc}/models/non-linear', 'contents': [ {{ 'type': 'file', 'name': 'PN_junction.h', 'external-contents': '{own_inc}/models/non-linear/PN_junction.h' }} ] }}\n"
                "] }\n")

    def build(rel):
        exe = os.path.join(OUT, rel.replace("/", "__").replace(".cpp", ""))
        extra = ["-ivfsoverlay", vfs, "-idirafter", "/root/reference/include"] if "bsim3v32" in rel else []
        cmd = [CXX, "-std=c++23", "-O1", "-w", f"-I{ROOT}/phy-engine_amd/include", f"-I{ROOT}/include"] + extra + ["-o", exe, os.path.join(REF_TESTS, rel),
               f"-L{EMU}", "-lpe_hip_emu", f"-Wl,-rpath,{EMU}"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return rel, exe, r.returncode, r.stderr[-2000:]

    with ThreadPoolExecutor(max_workers=7) as pool:
        return {rel: (exe, rc, err) for rel, exe, rc, err in pool.map(build, _programs())}


@pytest.mark.parametrize("rel", _programs())
def test_reference_program_compiles_unchanged_and_passes(built, rel):
    exe, rc, err = built[rel]
    assert rc == 0, f"{rel} does not compile against phy-engine_amd/include:\n{err}"
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    want = EXPECTED.get(rel, 0)
    assert run.returncode == want, f"{rel} exited {run.returncode} (expected {want}):\n{run.stderr[-1500:]}"
    if rel == "0005.models/cutthrough.cpp":
        assert "R1: VA=(3.000000,0.000000), VB=(3.000000,0.000000)" in run.stdout  # first circuit (r = DBL_MIN): as the reference prints it


def test_reference_program_list_is_complete():
    names = _programs()
    assert len(names) >= 140 and "0001.module/concept.cpp" in names and "0002.net_list/operation.cpp" in names
    assert sum("bsim3v32" in n for n in names) == len(glob.glob(os.path.join(REF_TESTS, "0004.solver", "bsim3v32_*.cpp"))) + len(glob.glob(os.path.join(REF_TESTS, "0012.ac", "bsim3v32_*.cpp")))
    assert sum(n.startswith("0005.models/") for n in names) == len(glob.glob(os.path.join(REF_TESTS, "0005.models", "*.cpp")))
