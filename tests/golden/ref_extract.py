"""Runs the reference's OWN numpy path (build container only) so that the oracle and the golden
fields are pinned to the reference's text rather than to a restatement of it.

The vectorised numpy path of /root/reference/latticeboltzmann.py is commented out there and the
module cannot be imported (top-level `import pycuda.autoinit`, :11).  This script READS that file at
run time -- it never stores its text -- takes

    :16-35    dtype, rtol, D2Q9 constants and their four asserts             (live code)
    :80-84    OMEGA, p_ambient, u_ambient, p_insides, u_insides              (live code)
    :87-95    getEquilibrium, surroundings, insides and the two asserts      (live code)
    :98       omega                                                          (commented out)
    :107-123  collide, stream, reflect                                       (commented out)
    :142      initial cells                                                  (live code)
    :180-183  moments: p, u, nan_to_num                                      (commented out)

verbatim (the commented-out ones with their leading `# ` removed), executes them with numpy for a
given (N, M, blocked), and steps them in the order of the only surviving copy of the loop,
/root/reference/latticeboltzmann.jl:107-120 (moments, collide, [display], stream, reflect).

Nothing under tests/ (other than this generator and the skip-if-absent live cross-check in
tests/test_oracle.py) reads /root/reference; the GPU box has no /root/reference at all.
"""
import math
import os
import re

import numpy as np

REF_PY = "/root/reference/latticeboltzmann.py"


def available():
    return os.path.exists(REF_PY)


def _lines(src, lo, hi, commented=False):
    """Lines lo..hi (1-based, inclusive) of the reference file as one code string.  commented=True
    removes the comment marker the reference put in front of each line (the common indentation,
    `#` and one blank)."""
    out = []
    for ln in src[lo - 1:hi]:
        if commented:
            ln = re.sub(r"^\s*# ?", "", ln, count=1)
        out.append(ln)
    return "\n".join(out) + "\n"


class ReferenceNumpyPath:
    """The reference's numpy path, executed from its own text.

    overrides: values for OMEGA / p_ambient / u_ambient (the reference edits them in the source,
    :80-82; latticeboltzmann.jl:27-29 is the other parameter set it names).
    stream: optional replacement for the reference's stream(cells) -- only for the periodic
    scenes, which have no reference counterpart.
    """

    def __init__(self, blocked, overrides=None, cells=None, stream=None):
        with open(REF_PY) as f:
            src = f.read().splitlines()
        blocked = np.ascontiguousarray(blocked, dtype=bool)
        ns = {"np": np, "math": math, "blocked": blocked, "N": blocked.shape[0], "M": blocked.shape[1]}
        exec(compile(_lines(src, 16, 35), REF_PY + ":16-35", "exec"), ns)
        exec(compile(_lines(src, 80, 82), REF_PY + ":80-82", "exec"), ns)
        for k, v in (overrides or {}).items():
            assert k in ("OMEGA", "p_ambient", "u_ambient"), k
            ns[k] = v
        exec(compile(_lines(src, 83, 84), REF_PY + ":83-84", "exec"), ns)
        exec(compile(_lines(src, 86, 95), REF_PY + ":86-95", "exec"), ns)
        exec(compile(_lines(src, 98, 98, commented=True), REF_PY + ":98", "exec"), ns)
        exec(compile(_lines(src, 107, 123, commented=True), REF_PY + ":107-123", "exec"), ns)
        exec(compile(_lines(src, 142, 142), REF_PY + ":142", "exec"), ns)
        self._moments = compile(_lines(src, 180, 183, commented=True), REF_PY + ":180-183", "exec")
        if cells is not None:
            ns["cells"] = np.array(cells, dtype=ns["dtype"], copy=True)
        if stream is not None:
            ns["stream"] = stream
        self.ns = ns

    @property
    def cells(self):
        return self.ns["cells"]

    @property
    def surroundings(self):
        return self.ns["surroundings"]

    @property
    def omega(self):
        return self.ns["omega"]

    def moments(self):
        """(u, p) of the current cells, by the reference's own lines :180-183."""
        exec(self._moments, self.ns)
        return self.ns["u"], self.ns["p"]

    def step(self, n=1):
        ns = self.ns
        for _ in range(n):  # latticeboltzmann.jl:107-120
            exec(self._moments, ns)
            ns["collide"](ns["cells"], ns["u"], ns["p"])
            ns["stream"](ns["cells"])
            ns["reflect"](ns["cells"])
        return ns["cells"]
