"""On-disk point-cloud format of the reference: ``save_ply`` / ``load_ply`` of
``src/scene/gaussian_model.py:458-579``.

``save_ply(path)`` writes two files like the reference: ``raw_<name>`` with every attribute (``label_0`` included --
this is the one ``load_ply`` reads back) and ``<name>`` without the label column ("a hack to not re-write the
visualization software", :507-514).  The reference goes through the ``plyfile`` package; here the (tiny) binary
little-endian PLY container is written and parsed directly with numpy -- same header text, property order and float32
payload as ``PlyElement.describe(...)`` + ``PlyData.write`` produce.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}


def write_ply_vertices(path: str, names, columns: np.ndarray) -> None:
    """Binary little-endian PLY with one ``vertex`` element of float properties ``names`` (columns: [N, len(names)])."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    assert columns.ndim == 2 and columns.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", "element vertex %d" % columns.shape[0]]
    header += ["property float %s" % n for n in names]
    header.append("end_header")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def read_ply_vertices(path: str):
    """Returns (names, structured array) of the first element of a PLY file (binary little-endian or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, count, props, in_first, seen_element = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_first = not seen_element
                seen_element = True
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        dtype = np.dtype(props)
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype, count=count)
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            data = np.zeros(count, dtype=dtype)
            for i, (n, _) in enumerate(props):
                data[n] = rows[:, i]
        else:
            raise ValueError("unsupported PLY format %r" % fmt)
    return [n for n, _ in props], data


class PlyMixin:
    def construct_list_of_attributes(self, remove_label=False):
        """gaussian_model.py:458-477."""
        names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
        names += ['f_dc_%d' % i for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += ['f_rest_%d' % i for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ['opacity', 'orient_conf']
        if not remove_label:
            names.append('label_0')
        names += ['scale_%d' % i for i in range(self._scaling.shape[1])]
        names += ['rot_%d' % i for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        """gaussian_model.py:479-514."""
        d, name = os.path.dirname(path), os.path.basename(path)

        def npy(t):
            return t.detach().cpu().numpy()

        xyz = npy(self._xyz)
        normals = np.zeros_like(xyz)
        f_dc = npy(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        f_rest = npy(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        opac, conf, lab = npy(self._opacity), npy(self._orient_conf), npy(self._label)
        scale, rot = npy(self._scaling), npy(self._rotation)
        write_ply_vertices(os.path.join(d, "raw_" + name), self.construct_list_of_attributes(),
                           np.concatenate((xyz, normals, f_dc, f_rest, opac, conf, lab, scale, rot), axis=1))
        write_ply_vertices(path, self.construct_list_of_attributes(remove_label=True),
                           np.concatenate((xyz, normals, f_dc, f_rest, opac, conf, scale, rot), axis=1))

    def load_ply(self, path, device=None):
        """gaussian_model.py:520-579 (reads the ``raw_`` variant: ``label_0`` is required, ``orient_conf`` optional)."""
        names, v = read_ply_vertices(path)
        dev = device if device is not None else (self._xyz.device if self._xyz.numel() else "cpu")
        n = v.shape[0]
        xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
        opac = np.asarray(v["opacity"])[..., None]
        conf = np.asarray(v["orient_conf"])[..., None] if "orient_conf" in names else np.zeros((n, 1))
        labels = np.asarray(v["label_0"])[..., None]
        f_dc = np.zeros((n, 3, 1))
        for c in range(3):
            f_dc[:, c, 0] = v["f_dc_%d" % c]
        extra = sorted([p for p in names if p.startswith("f_rest_")], key=lambda x: int(x.split('_')[-1]))
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_extra = np.stack([v[p] for p in extra], axis=1).reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)
        sc = sorted([p for p in names if p.startswith("scale_")], key=lambda x: int(x.split('_')[-1]))
        ro = sorted([p for p in names if p.startswith("rot")], key=lambda x: int(x.split('_')[-1]))

        def par(a):
            return nn.Parameter(torch.tensor(np.asarray(a), dtype=torch.float, device=dev).requires_grad_(True))

        self._xyz = par(xyz)
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=dev).transpose(1, 2).contiguous()
                                         .requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_extra, dtype=torch.float, device=dev).transpose(1, 2)
                                           .contiguous().requires_grad_(True))
        self._opacity, self._orient_conf, self._label = par(opac), par(conf), par(labels)
        if sc:
            self._scaling = par(np.stack([v[p] for p in sc], axis=1))
        if ro:
            self._rotation = par(np.stack([v[p] for p in ro], axis=1))
        self.max_radii2D = torch.zeros(n, device=dev)
        self.active_sh_degree = self.max_sh_degree
