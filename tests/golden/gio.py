"""bohip-golden-v1: a line-oriented text format both Python and a dependency-free Julia script can read and write
(julia/gen_golden.jl has no JSON/NPZ package to rely on).  Values are written with repr() = shortest round-trip
decimal, so float64 survives bit for bit.

    bohip-golden-v1
    case <name>
    str <key> <value>
    array <key> <ndim> <dim...>        (row-major; a scalar is ndim 0)
    <values, whitespace separated, on one line>
    end
"""
import numpy as np

MAGIC = "bohip-golden-v1"


def write_cases(path, cases):
    with open(path, "w") as f:
        f.write(MAGIC + "\n")
        for name, fields in cases.items():
            f.write(f"case {name}\n")
            for k, v in fields.items():
                if isinstance(v, str):
                    f.write(f"str {k} {v}\n")
                    continue
                a = np.asarray(v, dtype=np.float64)
                f.write(f"array {k} {a.ndim} {' '.join(str(s) for s in a.shape)}".rstrip() + "\n")
                f.write(" ".join(repr(float(x)) for x in a.ravel()) + "\n")
            f.write("end\n")


def read_cases(path):
    cases, cur = {}, None
    with open(path) as f:
        assert f.readline().strip() == MAGIC, "not a bohip-golden-v1 file"
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "case":
                cur = cases.setdefault(t[1], {})
            elif t[0] == "str":
                cur[t[1]] = " ".join(t[2:])
            elif t[0] == "array":
                nd = int(t[2])
                shape = tuple(int(s) for s in t[3:3 + nd])
                vals = np.array([float(x) for x in f.readline().split()], dtype=np.float64)
                cur[t[1]] = vals.reshape(shape) if nd else vals.reshape(())
            elif t[0] == "end":
                cur = None
    return cases
