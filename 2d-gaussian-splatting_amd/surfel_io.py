"""On-disk formats of the reference without third-party readers: `point_cloud.ply` as
scene/gaussian_model.py:193-207 writes it through plyfile (binary little-endian, one `vertex` element, float32
properties x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*) and as :214-255 reads it back.  numpy only.
"""
import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def write_ply(path, names, columns):
    """columns: [P, len(names)] float32 -> binary_little_endian PLY with one float property per name."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    if columns.ndim != 2 or columns.shape[1] != len(names):
        raise ValueError("columns must be [P, %d]" % len(names))
    header = ["ply", "format binary_little_endian 1.0", "element vertex %d" % columns.shape[0]]
    header += ["property float %s" % n for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def read_ply(path):
    """Returns {property name: 1-D numpy array} of the first element (`vertex`).  Handles binary little/big endian and
    ascii; list properties are not supported (the reference's point clouds have none)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, count, props, in_first, n_elements = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                n_elements += 1
                in_first = n_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("%s: list properties are not supported" % path)
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError("%s: PLY header lacks format / element" % path)
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, i].astype(t) for i, (n, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: np.ascontiguousarray(rec[n]) for n, _ in props}
