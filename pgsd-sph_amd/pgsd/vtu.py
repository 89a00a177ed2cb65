"""Particle frames as VTK XML unstructured-grid files (``.vtu``) for ParaView / VisIt.

The reference ships only the beginning of such a converter (``test_pgsd2vtu.py:20-30`` opens the
trajectory and walks the frames; the ``pyevtk`` call that would write them is missing). This
module writes the files itself, with no dependency beyond numpy: one ``VTK_VERTEX`` cell per
particle, every per-particle array of the frame as point data, raw appended binary blocks
(little endian, 64-bit block headers).
"""
import os
import struct

import numpy

_VTK_TYPES = {
    'int8': 'Int8', 'uint8': 'UInt8', 'int16': 'Int16', 'uint16': 'UInt16', 'int32': 'Int32',
    'uint32': 'UInt32', 'int64': 'Int64', 'uint64': 'UInt64', 'float32': 'Float32', 'float64': 'Float64',
}

#: per-particle attributes of ``pgsd.hoomd.ParticleData`` written as point data when present
POINT_FIELDS = ('typeid', 'mass', 'body', 'velocity', 'slength', 'density', 'pressure', 'energy',
                'auxiliary1', 'auxiliary2', 'auxiliary3', 'auxiliary4', 'image',
                'charge', 'diameter', 'orientation', 'angmom', 'moment_inertia')


def _host(a):
    if hasattr(a, 'detach'):  # torch tensor (device frames)
        a = a.detach().cpu().numpy()
    return numpy.ascontiguousarray(a)


def write_vtu(path, points, point_data=None, field_data=None):
    """Write N points and their arrays as an UnstructuredGrid of N vertex cells.

    Args:
        path (str): output file name.
        points: ``N x 3`` array.
        point_data (dict): name -> array with N rows (1, 3 or more components).
        field_data (dict): name -> small array stored once per file (step, box).
    """
    points = _host(points)
    if points.ndim != 2 or points.shape[1] != 3:
        raise ValueError("points must be an N x 3 array")
    if points.dtype not in (numpy.float32, numpy.float64):
        points = points.astype(numpy.float32)
    n = points.shape[0]
    blocks = []
    offset = [0]

    def data_array(name, arr, ncomp, indent, tuples=None):
        arr = numpy.ascontiguousarray(arr)
        vt = _VTK_TYPES.get(arr.dtype.name)
        if vt is None:
            raise ValueError("no VTK type for dtype %s (%s)" % (arr.dtype, name))
        arr = arr.astype(arr.dtype.newbyteorder('<'), copy=False)
        extra = '' if tuples is None else ' NumberOfTuples="%d"' % tuples
        line = '%s<DataArray type="%s" Name="%s" NumberOfComponents="%d"%s format="appended" offset="%d"/>\n' \
            % (indent, vt, name, ncomp, extra, offset[0])
        blocks.append(arr)
        offset[0] += 8 + arr.nbytes
        return line

    xml = ['<?xml version="1.0"?>\n',
           '<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n',
           '  <UnstructuredGrid>\n']
    if field_data:
        xml.append('    <FieldData>\n')
        for name, arr in field_data.items():
            arr = numpy.atleast_1d(_host(arr)).reshape(-1)
            xml.append(data_array(name, arr, 1, '      ', tuples=arr.size))
        xml.append('    </FieldData>\n')
    xml.append('    <Piece NumberOfPoints="%d" NumberOfCells="%d">\n' % (n, n))
    xml.append('      <Points>\n')
    xml.append(data_array('points', points, 3, '        '))
    xml.append('      </Points>\n')
    xml.append('      <Cells>\n')
    xml.append(data_array('connectivity', numpy.arange(n, dtype=numpy.int64), 1, '        '))
    xml.append(data_array('offsets', numpy.arange(1, n + 1, dtype=numpy.int64), 1, '        '))
    xml.append(data_array('types', numpy.ones(n, dtype=numpy.uint8), 1, '        '))  # VTK_VERTEX
    xml.append('      </Cells>\n')
    xml.append('      <PointData>\n')
    for name, arr in (point_data or {}).items():
        arr = _host(arr)
        if arr.shape[0] != n:
            raise ValueError("point data %r has %d rows, expected %d" % (name, arr.shape[0], n))
        ncomp = 1 if arr.ndim == 1 else int(numpy.prod(arr.shape[1:]))
        xml.append(data_array(name, arr.reshape(n, ncomp), ncomp, '        '))
    xml.append('      </PointData>\n')
    xml.append('    </Piece>\n')
    xml.append('  </UnstructuredGrid>\n')
    xml.append('  <AppendedData encoding="raw">\n_')
    with open(path, 'wb') as f:
        f.write(''.join(xml).encode('ascii'))
        for arr in blocks:
            f.write(struct.pack('<Q', arr.nbytes))
            f.write(arr.tobytes())
        f.write(b'\n  </AppendedData>\n</VTKFile>\n')


def read_vtu_arrays(path):
    """Minimal reader of files written by :func:`write_vtu` (for checks): name -> array."""
    import re
    raw = open(path, 'rb').read()
    head, _, tail = raw.partition(b'<AppendedData encoding="raw">\n_')
    inverse = {v: k for k, v in _VTK_TYPES.items()}
    out = {}
    for m in re.finditer(rb'<DataArray type="(\w+)" Name="([^"]+)" NumberOfComponents="(\d+)"[^>]*offset="(\d+)"/>', head):
        dt = numpy.dtype(inverse[m.group(1).decode()]).newbyteorder('<')
        ncomp, off = int(m.group(3)), int(m.group(4))
        nbytes = struct.unpack_from('<Q', tail, off)[0]
        arr = numpy.frombuffer(tail, dtype=dt, count=nbytes // dt.itemsize, offset=off + 8)
        out[m.group(2).decode()] = arr.reshape(-1, ncomp) if ncomp > 1 else arr
    return out


def frame_to_vtu(path, frame):
    """One ``pgsd.hoomd.Frame`` (host or device arrays) -> one ``.vtu`` file."""
    p = frame.particles
    if p.position is None:
        raise ValueError("frame has no particle positions")
    data = {}
    for name in POINT_FIELDS:
        arr = getattr(p, name, None)
        if arr is not None and _host(arr).shape[:1] == _host(p.position).shape[:1]:
            data[name] = arr
    fields = {'step': numpy.array([frame.configuration.step or 0], dtype=numpy.uint64)}
    if frame.configuration.box is not None:
        fields['box'] = numpy.asarray(frame.configuration.box, dtype=numpy.float32)
    write_vtu(path, p.position, data, fields)


def pgsd2vtu(gsd_name, out_dir=None, frames=None):
    """Convert every frame (or ``frames``, an iterable of indices) of a trajectory.

    Files are named ``<stem>_<step, 9 digits>.vtu``; a ``<stem>.pvd`` collection lists them with
    their time step. Returns the list of files written."""
    from . import hoomd
    stem = os.path.splitext(os.path.basename(gsd_name))[0]
    out_dir = out_dir or os.path.dirname(os.path.abspath(gsd_name))
    os.makedirs(out_dir, exist_ok=True)
    written = []
    with hoomd.open(gsd_name, 'r') as traj:
        indices = range(len(traj)) if frames is None else frames
        for i in indices:
            frame = traj[i]
            step = int(frame.configuration.step or 0)
            name = os.path.join(out_dir, "%s_%09d.vtu" % (stem, step))
            frame_to_vtu(name, frame)
            written.append((step, name))
    with open(os.path.join(out_dir, stem + ".pvd"), 'w') as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n  <Collection>\n')
        for step, name in written:
            f.write('    <DataSet timestep="%d" part="0" file="%s"/>\n' % (step, os.path.basename(name)))
        f.write('  </Collection>\n</VTKFile>\n')
    return [name for _, name in written]
