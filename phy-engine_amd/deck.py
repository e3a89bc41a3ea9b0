"""pe-deck: the plain netlist description (text format + in-memory form) from which the HIP engine's device
tables are built; the test infrastructure reads the same text format.

A deck is topology + per-device parameters, in *model insertion order* (the order fixes MNA row
numbering exactly as the reference's `circult::prepare()` does, circuits/circuit.h:481-540):

    nodes <N>                       # analog nodes 1..N, created in this order; 0 is ground
    R   a b  r                      # model/models/linear/resistance.h
    C   a b  c                      # model/models/linear/capacitor.h
    L   a b  l                      # model/models/linear/inductor.h        (1 branch)
    VDC a b  V                      # model/models/linear/VDC.h             (1 branch)
    VAC a b  Vp omega phase         # model/models/linear/VAC.h             (1 branch)
    IDC a b  I                      # model/models/linear/IDC.h
    D   a c  Is N Isr Nr Temp Ibv Bv Bv_set Area tt   # non-linear/PN_junction.h
    FBR a b p m                     # non-linear/full_bridge_rectifier.h (4 default diodes)
    IAC  a b  Ip omega phase        # linear/IAC.h
    VCCS s t p q  g                 # linear/VCCS.h
    VCVS s t p q  mu                # linear/VCVS.h                         (1 branch)
    CCCS s t p q  alpha             # linear/CCCS.h                         (1 branch)
    CCVS s t p q  r                 # linear/CCVS.h                         (2 branches)
    OPAMP s t p q  mu               # linear/op_amp.h                       (1 branch)
    XFMR p q s t  n                 # linear/transformer.h                  (2 branches)
    SW   a b  cut_through           # controller/switch.h                   (1 branch)
    SAW  p m  Vh Vl freq phase      # generator/sawtooth.h                  (1 branch)
    SQR  p m  Vh Vl freq duty phase # generator/square.h                    (1 branch)
    PULSE p m Vh Vl freq duty phase tr tf   # generator/pulse.h             (1 branch)
    TRI  p m  Vh Vl freq phase      # generator/triangle.h                  (1 branch)
    KL   p1 p2 s1 s2  L1 L2 k       # linear/coupled_inductors.h            (2 branches)
    NMOS d g s  Kp lambda Vth       # non-linear/nmosfet.h   (level 1)
    PMOS d g s  Kp lambda Vth       # non-linear/pmosfet.h
    NPN  b c e  Is N BetaF Temp Area   # non-linear/BJT_NPN.h
    PNP  b c e  Is N BetaF Temp Area   # non-linear/BJT_PNP.h
    RELAY cp cn a b  Von Voff       # controller/relay.h                    (1 branch)
    XCT  p q s1 ct s2  n_total      # linear/transformer_center_tap.h       (3 branches)

Node id -1 = unconnected pin.  Values are printed with %.17g so every consumer reads the same doubles.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

KINDS = ("R", "C", "L", "VDC", "VAC", "IDC", "D", "FBR", "IAC", "VCCS", "VCVS", "CCCS", "CCVS", "OPAMP", "XFMR", "SW", "SAW", "SQR", "PULSE",
         "TRI", "KL", "NMOS", "PMOS", "NPN", "PNP", "RELAY", "XCT")
NPINS = {"R": 2, "C": 2, "L": 2, "VDC": 2, "VAC": 2, "IDC": 2, "D": 2, "FBR": 4, "IAC": 2, "VCCS": 4, "VCVS": 4, "CCCS": 4, "CCVS": 4,
         "OPAMP": 4, "XFMR": 4, "SW": 2, "SAW": 2, "SQR": 2, "PULSE": 2, "TRI": 2, "KL": 4, "NMOS": 3, "PMOS": 3, "NPN": 3, "PNP": 3, "RELAY": 4, "XCT": 5}
NBRANCH = {"R": 0, "C": 0, "L": 1, "VDC": 1, "VAC": 1, "IDC": 0, "D": 0, "FBR": 0, "IAC": 0, "VCCS": 0, "VCVS": 1, "CCCS": 1, "CCVS": 2,
           "OPAMP": 1, "XFMR": 2, "SW": 1, "SAW": 1, "SQR": 1, "PULSE": 1, "TRI": 1, "KL": 2, "NMOS": 0, "PMOS": 0, "NPN": 0, "PNP": 0, "RELAY": 1, "XCT": 3}
# defaults follow the reference structs' member initialisers
DEFAULTS = {
    "R": (10.0,),
    "C": (1e-5,),
    "L": (1e-5,),
    "VDC": (5.0,),
    "VAC": (5.0, 50.0, 0.0),
    "IDC": (1.0,),
    # Is N Isr Nr Temp Ibv Bv Bv_set Area tt   (PN_junction.h:26-37)
    "D": (1e-14, 1.0, 0.0, 2.0, 27.0, 1e-3, 40.0, 1.0, 1.0, 0.0),
    "FBR": (),
    "IAC": (1.0, 50.0, 0.0),
    "VCCS": (1.0,), "VCVS": (1.0,), "CCCS": (1.0,), "CCVS": (1.0,), "OPAMP": (1e5,), "XFMR": (1.0,),
    "SW": (0.0,),
    "SAW": (5.0, 0.0, 1e3, 0.0),
    "SQR": (5.0, 0.0, 1e3, 0.5, 0.0),
    "PULSE": (5.0, 0.0, 1e3, 0.5, 0.0, 0.0, 0.0),
    "TRI": (5.0, 0.0, 1e3, 0.0),
    "KL": (1e-3, 1e-3, 0.99),
    "NMOS": (1e-3, 0.0, 1.0), "PMOS": (1e-3, 0.0, 1.0),                   # nmosfet.h:19-21
    "NPN": (1e-16, 1.0, 100.0, 27.0, 1.0), "PNP": (1e-16, 1.0, 100.0, 27.0, 1.0),   # BJT_NPN.h:15-19
    "RELAY": (5.0, 3.0), "XCT": (1.0,),
}
# generators -> (type code of PE_HIP_VGEN, positions of Vh Vl freq duty phase tr tf in the deck's parameter tuple, -1 = absent)
VGEN_LAYOUT = {"SAW": (0, (0, 1, 2, -1, 3, -1, -1)), "SQR": (1, (0, 1, 2, 3, 4, -1, -1)), "PULSE": (2, (0, 1, 2, 3, 4, 5, 6)),
               "TRI": (3, (0, 1, 2, -1, 3, -1, -1))}


@dataclass
class Deck:
    n_nodes: int = 0
    devices: list = field(default_factory=list)  # (kind, nodes tuple, params tuple)

    def add(self, kind: str, nodes, *params):
        assert kind in KINDS and len(nodes) == NPINS[kind]
        d = DEFAULTS[kind]
        p = tuple(float(x) for x in params) + d[len(params):]
        self.devices.append((kind, tuple(int(n) for n in nodes), p))
        return len(self.devices) - 1

    def new_node(self) -> int:
        self.n_nodes += 1
        return self.n_nodes

    # ---- row numbering (circuit.h:481-540): analog nodes in creation order, then branches in model order
    @property
    def n_branches(self) -> int:
        return sum(NBRANCH[k] for k, _, _ in self.devices)

    @property
    def rows(self) -> int:
        return self.n_nodes + self.n_branches

    def count(self, kind: str) -> int:
        return sum(1 for k, _, _ in self.devices if k == kind)

    def has_nonlinear(self) -> bool:
        return any(k in ("D", "FBR", "NMOS", "PMOS", "NPN", "PNP", "RELAY") for k, _, _ in self.devices)

    def dumps(self) -> str:
        out = [f"nodes {self.n_nodes}"]
        for k, n, p in self.devices:
            out.append(" ".join([k] + [str(x) for x in n] + ["%.17g" % x for x in p]))
        return "\n".join(out) + "\n"

    def write(self, path):
        with open(path, "w") as f:
            f.write(self.dumps())

    @staticmethod
    def loads(text: str) -> "Deck":
        d = Deck()
        for line in text.splitlines():
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            tok = line.split()
            if tok[0] == "nodes":
                d.n_nodes = int(tok[1])
                continue
            k = tok[0]
            np_ = NPINS[k]
            d.add(k, [int(x) for x in tok[1:1 + np_]], *[float(x) for x in tok[1 + np_:]])
        return d

    @staticmethod
    def read(path) -> "Deck":
        with open(path) as f:
            return Deck.loads(f.read())


# --------------------------------------------------------------------------------------------
# portable generator (BASELINE.md §3 "Inputs"): splitmix64 + Box-Muller.  The reference's own
# benchmarks use OS entropy (benchmark/series_parallel.cpp:8), so seeds are ours.
# --------------------------------------------------------------------------------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int) -> np.ndarray:
    """n doubles in (0,1), 53-bit."""
    return ((splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normals(seed: int, n: int) -> np.ndarray:
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    th = 2.0 * math.pi * u[1::2]
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(th)
    z[1::2] = r * np.sin(th)
    return z[:n]


# --------------------------------------------------------------------------------------------
# workloads of BASELINE.json / SURVEY.md §8(d)
# --------------------------------------------------------------------------------------------
def rc_mesh(W: int = 100, H: int = 100, seed: int = 1, nonlinear: bool = False, jitter: float = 0.05) -> Deck:
    """C3: W x H RC mesh.  node(i,j) = i*W + j + 1, source node s = W*H + 1.
    Insertion order: for i, for j: [R right], [R down], C, [D if (i+j)%8==0 and nonlinear]; then V, R_s.
    One normal draw per jittered device, in insertion order (R right, R down, C)."""
    d = Deck()
    d.n_nodes = W * H + 1
    s = W * H + 1
    z = normals(seed, 3 * W * H)
    zi = 0

    def node(i, j):
        return i * W + j + 1

    for i in range(H):
        for j in range(W):
            if j + 1 < W:
                d.add("R", (node(i, j), node(i, j + 1)), 1000.0 * (1.0 + jitter * z[zi]))
            zi += 1
            if i + 1 < H:
                d.add("R", (node(i, j), node(i + 1, j)), 1000.0 * (1.0 + jitter * z[zi]))
            zi += 1
            d.add("C", (node(i, j), 0), 1e-12 * (1.0 + jitter * z[zi]))
            zi += 1
            if nonlinear and (i + j) % 8 == 0:
                d.add("D", (node(i, j), 0))
    if nonlinear:
        d.add("VAC", (s, 0), 2.0, 2.0 * math.pi * 1e8, 0.0)
    else:
        d.add("VDC", (s, 0), 1.0)
    d.add("R", (s, node(0, 0)), 50.0)
    return d


def rc_mesh_params(W: int, H: int, seeds, nonlinear: bool = False, jitter: float = 0.05):
    """Per-instance parameter vectors for the Monte-Carlo sweep (C5): same topology as rc_mesh(),
    instance k uses seeds[k].  Returns (deck of seeds[0], r[B][nR], c[B][nC])."""
    base = rc_mesh(W, H, seeds[0], nonlinear, jitter)
    nR, nC = base.count("R"), base.count("C")
    r = np.empty((len(seeds), nR))
    c = np.empty((len(seeds), nC))
    # mask of which draws are used, in insertion order
    use_r = np.zeros(3 * W * H, dtype=bool)
    use_c = np.zeros(3 * W * H, dtype=bool)
    k = 0
    for i in range(H):
        for j in range(W):
            use_r[k] = j + 1 < W
            use_r[k + 1] = i + 1 < H
            use_c[k + 2] = True
            k += 3
    for b, sd in enumerate(seeds):
        z = normals(int(sd), 3 * W * H)
        r[b, :nR - 1] = 1000.0 * (1.0 + jitter * z[use_r])
        r[b, nR - 1] = 50.0
        c[b, :] = 1e-12 * (1.0 + jitter * z[use_c])
    return base, r, c


def bridge_rectifier() -> Deck:
    """C2: VAC 10 V / 50 Hz -> full_bridge_rectifier -> 1 kOhm || 100 uF."""
    d = Deck()
    d.n_nodes = 3
    d.add("VAC", (1, 2), 10.0, 2.0 * math.pi * 50.0, 0.0)
    d.add("FBR", (1, 2, 3, 0))
    d.add("R", (3, 0), 1000.0)
    d.add("C", (3, 0), 100e-6)
    return d


def resistor_ladder(n: int = 1000, merges: int = 100, seed: int = 1) -> Deck:
    """C1: n-resistor chain, r ~ U(1e-5,1e5), `merges` random node merges, VDC 3 V, DC.
    Merging is done on the topology (union-find) before the deck is written."""
    u = uniform01(seed, n + 2 * merges)
    r = 1e-5 + (1e5 - 1e-5) * u[:n]
    parent = list(range(n + 1))  # chain nodes 0..n ; 0 -> VDC+, n -> ground

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for m in range(merges):
        a = int(u[n + 2 * m] * (n - 1)) + 1
        b = int(u[n + 2 * m + 1] * (n - 1)) + 1
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    gnd = find(n)
    ids = {}
    d = Deck()

    def nid(a):
        ra = find(a)
        if ra == gnd:
            return 0
        if ra not in ids:
            ids[ra] = d.new_node()
        return ids[ra]

    top = nid(0)
    for k in range(n):
        a, b = nid(k), nid(k + 1)
        d.add("R", (a, b), r[k])
    d.add("VDC", (top, 0), 3.0)
    return d


def rc_step() -> Deck:
    """test/0005.models/rc_step_tr.cpp: VDC 1 V - R 1k - C 1n, dt = tau/100."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (1, 2), 1000.0)
    d.add("C", (2, 0), 1e-9)
    return d


def rl_step() -> Deck:
    """test/0005.models/rl_step_tr.cpp shape: VDC - R - L to ground."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (1, 2), 1000.0)
    d.add("L", (2, 0), 1e-3)
    return d


def diode_op() -> Deck:
    """test/0011.nonlinear/op_pn_junction.cpp: VDC 1 V - R 1k - default PN to ground."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (1, 2), 1000.0)
    d.add("D", (2, 0))
    return d


def divider_dc() -> Deck:
    """test/0004.solver/dc.cpp: R1 10 (node2-node1), R2 20 (node1-gnd), VDC 3 V (node2-gnd) -> 3/2 V, 0.1 A."""
    d = Deck()
    d.n_nodes = 2
    d.add("R", (2, 1), 10.0)
    d.add("R", (1, 0), 20.0)
    d.add("VDC", (2, 0), 3.0)
    return d


def rlc_series_vl() -> Deck:
    """V source in series with L (node between carries no conductance -> zero MNA diagonal)."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 1.0)
    d.add("L", (1, 2), 1e-3)
    d.add("R", (2, 3), 100.0)
    d.add("C", (3, 0), 1e-6)
    d.add("IDC", (0, 3), 1e-3)
    return d


def floating_rc() -> Deck:
    """test/0003.circuits/operations.cpp shape: R || C with no ground reference -> singular."""
    d = Deck()
    d.n_nodes = 2
    d.add("R", (1, 2), 1000.0)
    d.add("C", (1, 2), 1e-6)
    return d


# --------------------------------------------------------------------------------------------
# decks of the reference's own model tests (test/0005.models/*.cpp) + transient variants
# --------------------------------------------------------------------------------------------
def vccs_dc() -> Deck:
    """test/0005.models/vccs_dc.cpp: VDC 5 V control, VCCS 2 mS sinking into 1 kOhm: vout = -g vctrl rload = -10 V."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 5.0)
    d.add("VCCS", (2, 0, 1, 0), 2e-3)
    d.add("R", (2, 0), 1000.0)
    return d


def vcvs_gain() -> Deck:
    """test/0005.models/vcvs_gain.cpp: mu = 2, Vin = 1 V: vout = 2 V."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 1.0)
    d.add("VCVS", (2, 0, 1, 0), 2.0)
    d.add("R", (2, 0), 1000.0)
    return d


def cccs_dc() -> Deck:
    """test/0005.models/cccs_dc.cpp: i_ctrl = 5 mA, alpha = 3: vout = -15 V."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 5.0)
    d.add("R", (1, 2), 1000.0)
    d.add("CCCS", (3, 0, 2, 0), 3.0)
    d.add("R", (3, 0), 1000.0)
    return d


def ccvs_dc() -> Deck:
    """test/0005.models/ccvs_dc.cpp: i_ctrl = 5 mA, r = 2 kOhm: vout = 10 V."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 5.0)
    d.add("R", (1, 2), 1000.0)
    d.add("CCVS", (3, 0, 2, 0), 2000.0)
    d.add("R", (3, 0), 1000.0)
    return d


def op_amp_follower() -> Deck:
    """test/0005.models/op_amp_follower.cpp: mu = 1e6 follower: vout = mu/(1+mu) Vin."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 1.0)
    d.add("OPAMP", (1, 2, 2, 0), 1e6)
    d.add("R", (2, 0), 1000.0)
    return d


def transformer_ratio() -> Deck:
    """test/0005.models/transformer_ratio.cpp: n = 2, 4 V primary: 2 V on the 100 Ohm secondary load, Is + n Ip = 0."""
    d = Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 4.0)
    d.add("XFMR", (1, 0, 2, 0), 2.0)
    d.add("R", (2, 0), 100.0)
    return d


def switch_divider(closed: bool = False) -> Deck:
    """test/0005.models/switch_r_open.cpp (open) / switch.cpp (closed): VDC 1 V - switch - 1 kOhm."""
    d = Deck()
    d.n_nodes = 2
    d.add("SW", (1, 2), 1.0 if closed else 0.0)
    d.add("R", (2, 0), 1000.0)
    d.add("VDC", (1, 0), 1.0)
    return d


def generator_dc() -> Deck:
    """test/0005.models/generator_dc.cpp: eight generators, values at t = 0."""
    d = Deck()
    d.n_nodes = 8
    Vh, Vl, f = 5.0, 1.0, 1000.0
    d.add("SQR", (1, 0), Vh, Vl, f, 0.25, 0.0)
    d.add("SQR", (2, 0), Vh, Vl, f, 0.25, math.pi)
    d.add("SAW", (3, 0), Vh, Vl, f, 0.0)
    d.add("SAW", (4, 0), Vh, Vl, f, math.pi)
    d.add("TRI", (5, 0), Vh, Vl, f, 0.0)
    d.add("TRI", (6, 0), Vh, Vl, f, math.pi)
    d.add("PULSE", (7, 0), Vh, Vl, f, 0.1, 0.0, 0.0, 0.0)
    d.add("PULSE", (8, 0), Vh, Vl, f, 0.1, 1.5 * math.pi, 0.0, 0.0)
    for n in range(1, 9):   # a load on every source so that the branch currents are not all zero
        d.add("R", (n, 0), 100.0 * n)
    return d


def generators_tr() -> Deck:
    """Four generators (1 kHz .. 4 kHz, pulse with 20 us edges) each driving its own R-C low-pass; TR dt 5 us."""
    d = Deck()
    d.n_nodes = 8
    d.add("SAW", (1, 0), 5.0, -1.0, 1e3, 0.3)
    d.add("SQR", (2, 0), 3.0, 0.5, 2e3, 0.3, 1.0)
    d.add("PULSE", (3, 0), 4.0, 0.0, 3e3, 0.5, 0.2, 2e-5, 3e-5)
    d.add("TRI", (4, 0), 2.0, -2.0, 4e3, 2.0)
    for n in range(1, 5):
        d.add("R", (n, n + 4), 1000.0)
        d.add("C", (n + 4, 0), 1e-7)
    return d


def iac_rc() -> Deck:
    """IAC 1 mA / 1 kHz into R || C (TR); nothing in DC."""
    d = Deck()
    d.n_nodes = 1
    d.add("IAC", (0, 1), 1e-3, 2.0 * math.pi * 1e3, 0.5)
    d.add("R", (1, 0), 1000.0)
    d.add("C", (1, 0), 1e-7)
    return d


def coupled_inductors_tr(k: float = 0.0) -> Deck:
    """test/0005.models/coupled_inductors_TR.cpp (k = 0) and a coupled variant: VDC 1 V - L1 - 10 Ohm, L2 - 10 Ohm."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (2, 0), 10.0)
    d.add("KL", (1, 2, 0, 3), 1e-3, 1e-3, k)
    d.add("R", (3, 0), 10.0)
    return d


def controlled_mix() -> Deck:
    """Every linear controlled source + transformer + closed/open switches + a diode in one transient deck."""
    d = Deck()
    d.n_nodes = 10
    d.add("VAC", (1, 0), 2.0, 2.0 * math.pi * 1e4, 0.0)
    d.add("R", (1, 2), 100.0)
    d.add("C", (2, 0), 1e-7)
    d.add("VCVS", (3, 0, 2, 0), 3.0)
    d.add("R", (3, 4), 500.0)
    d.add("D", (4, 0))
    d.add("VCCS", (5, 0, 4, 0), 1e-3)
    d.add("R", (5, 0), 2000.0)
    d.add("C", (5, 0), 2e-8)
    d.add("CCCS", (6, 0, 5, 7), 2.0)
    d.add("R", (7, 0), 300.0)
    d.add("R", (6, 0), 150.0)
    d.add("XFMR", (6, 0, 8, 0), 0.5)
    d.add("SW", (8, 9), 1.0)
    d.add("R", (9, 0), 50.0)
    d.add("SW", (9, 10), 0.0)
    d.add("R", (10, 0), 1e4)
    d.add("L", (8, 0), 1e-3)
    return d


def nmos_common_source(vg: float = 2.0) -> Deck:
    """NMOS (Kp 2 mA/V^2, lambda 0.02, Vth 1) with a 2 kOhm drain resistor from 5 V, gate at vg: cutoff / saturation / triode by vg."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 5.0)
    d.add("VDC", (2, 0), vg)
    d.add("R", (1, 3), 2000.0)
    d.add("NMOS", (3, 2, 0), 2e-3, 0.02, 1.0)
    return d


def cmos_inverter_tr() -> Deck:
    """CMOS inverter (PMOS to 3.3 V, NMOS to ground) driven by a 100 kHz pulse with 1 us edges into 1 nF || 10 kOhm."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 3.3)
    d.add("PULSE", (2, 0), 3.3, 0.0, 1e5, 0.5, 0.0, 1e-6, 1e-6)
    d.add("PMOS", (3, 2, 1), 1e-3, 0.05, 0.8)
    d.add("NMOS", (3, 2, 0), 2e-3, 0.05, 0.7)
    d.add("C", (3, 0), 1e-9)
    d.add("R", (3, 0), 1e4)
    return d


def bjt_common_emitter(pnp: bool = False) -> Deck:
    """NPN (or the mirrored PNP) common-emitter stage: 100 kOhm base resistor, 1 kOhm collector resistor, 5 V rail."""
    d = Deck()
    d.n_nodes = 3
    s = -1.0 if pnp else 1.0
    d.add("VDC", (1, 0), 5.0 * s)
    d.add("R", (1, 2), 1e5)
    d.add("R", (1, 3), 1e3)
    d.add("PNP" if pnp else "NPN", (2, 3, 0), 1e-16, 1.0, 100.0, 27.0, 1.0)
    return d


def bjt_amp_tr() -> Deck:
    """AC-coupled NPN amplifier: VAC 10 mV / 10 kHz through 1 uF into the biased base, transient."""
    d = Deck()
    d.n_nodes = 5
    d.add("VDC", (1, 0), 9.0)
    d.add("R", (1, 2), 4.7e5)
    d.add("R", (1, 3), 2.2e3)
    d.add("NPN", (2, 3, 0), 1e-15, 1.0, 150.0, 27.0, 1.0)
    d.add("VAC", (4, 0), 0.01, 2.0 * math.pi * 1e4, 0.0)
    d.add("C", (4, 2), 1e-6)
    d.add("C", (3, 5), 1e-7)
    d.add("R", (5, 0), 1e4)
    return d


def center_tap_ratio() -> Deck:
    """test/0005.models/transformer_center_tap_ratio.cpp: 4 V primary, n_total 2: +1 V / -1 V on the two 100 Ohm half loads."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 4.0)
    d.add("XCT", (1, 0, 2, 0, 3), 2.0)
    d.add("R", (2, 0), 100.0)
    d.add("R", (3, 0), 100.0)
    return d


def relay_ramp() -> Deck:
    """test/0005.models/relay_hysteresis.cpp as ONE transient: a 100 Hz triangle (0..8 V) drives the coil of a relay
    (Von 5, Voff 3) whose contact switches 1 V onto 100 Ohm: closes on the way up past 5 V, opens on the way down past 3 V."""
    d = Deck()
    d.n_nodes = 3
    d.add("VDC", (1, 0), 1.0)
    d.add("TRI", (3, 0), 8.0, 0.0, 100.0, 0.0)
    d.add("R", (2, 0), 100.0)
    d.add("RELAY", (3, 0, 1, 2), 5.0, 3.0)
    d.add("R", (3, 0), 1e4)
    return d


# --------------------------------------------------------------------------------------------
# small-signal AC decks
# --------------------------------------------------------------------------------------------
def ac_rc_lowpass() -> Deck:
    """test/0012.ac/ac_omega.cpp: VAC 1 V - 1 kOhm - 1 uF: |vout| = 1/sqrt(2) at omega = 1/(R C) = 1000 rad/s."""
    d = Deck()
    d.n_nodes = 2
    d.add("VAC", (1, 0), 1.0, 1000.0, 0.0)
    d.add("R", (1, 2), 1000.0)
    d.add("C", (2, 0), 1e-6)
    return d


def ac_rlc_diode() -> Deck:
    """Series L into R || C with a forward-biased tt-diode to a 0.7 V bias source (ACOP: diode geq and tt geq from the OP)."""
    d = Deck()
    d.n_nodes = 3
    d.add("VAC", (1, 0), 1.0, 1e4, 0.3)
    d.add("VDC", (3, 0), 0.7)
    d.add("L", (1, 2), 1e-3)
    d.add("R", (2, 0), 50.0)
    d.add("C", (2, 0), 1e-6)
    d.add("D", (3, 2), 1e-14, 1.0, 0.0, 2.0, 27.0, 1e-3, 40.0, 1.0, 1.0, 1e-8)
    return d


def ac_linear_mix() -> Deck:
    """IAC + coupled inductors + transformer + VCVS / VCCS / CCCS + closed and open switches, all linear."""
    d = Deck()
    d.n_nodes = 8
    d.add("IAC", (0, 1), 1e-3, 2.0 * math.pi * 1e3, 0.4)
    d.add("R", (1, 0), 1000.0)
    d.add("C", (1, 2), 1e-7)
    d.add("KL", (2, 0, 3, 0), 1e-3, 4e-3, 0.8)
    d.add("R", (3, 0), 200.0)
    d.add("VCVS", (4, 0, 3, 0), 2.5)
    d.add("R", (4, 5), 100.0)
    d.add("XFMR", (5, 0, 6, 0), 2.0)
    d.add("SW", (6, 7), 1.0)
    d.add("R", (7, 0), 75.0)
    d.add("C", (7, 0), 2e-7)
    d.add("VCCS", (8, 0, 7, 0), 1e-3)
    d.add("R", (8, 0), 1e3)
    d.add("SW", (8, 1), 0.0)
    return d


def ac_nmos_amp() -> Deck:
    """Common-source NMOS stage (saturation at the OP) driven through a coupling capacitor: gm / gds small-signal stamps."""
    d = Deck()
    d.n_nodes = 5
    d.add("VDC", (1, 0), 5.0)
    d.add("VDC", (2, 0), 2.0)
    d.add("R", (2, 3), 1e5)
    d.add("VAC", (4, 0), 0.01, 2.0 * math.pi * 1e4, 0.0)
    d.add("C", (4, 3), 1e-6)
    d.add("R", (1, 5), 2000.0)
    d.add("NMOS", (5, 3, 0), 2e-3, 0.02, 1.0)
    d.add("C", (5, 0), 1e-9)
    return d
