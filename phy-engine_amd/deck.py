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

Node id -1 = unconnected pin.  Values are printed with %.17g so every consumer reads the same doubles.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

KINDS = ("R", "C", "L", "VDC", "VAC", "IDC", "D", "FBR")
NPINS = {"R": 2, "C": 2, "L": 2, "VDC": 2, "VAC": 2, "IDC": 2, "D": 2, "FBR": 4}
NBRANCH = {"R": 0, "C": 0, "L": 1, "VDC": 1, "VAC": 1, "IDC": 0, "D": 0, "FBR": 0}
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
}


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
        return any(k in ("D", "FBR") for k, _, _ in self.devices)

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
