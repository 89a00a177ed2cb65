"""Read and write HOOMD / HOOMD-SPH schema PGSD files, MI355X-native.

Same surface as the reference's ``pgsd.hoomd`` (/root/reference/pgsd/pgsd/hoomd.py):
``open``, ``HOOMDTrajectory``, ``Frame`` (``ConfigurationData``, ``ParticleData``,
``ConstraintData``), ``read_log``; the same chunk names, dtypes, shapes and frame-0 /
default-value fallbacks on read (hoomd.py:724-902).

The reference's writer is disabled (``append`` raises NotImplementedError, hoomd.py:568; its
intended policy survives only as comments, hoomd.py:569-642).  Here ``append`` is real and
follows that policy: per-particle arrays are partitioned over the ranks
(``write_all=True, offset=part_dist``), configuration scalars, ``N``, ``types`` and
``type_shapes`` are replicated small chunks (``write_all=False``).  Per-particle attributes may
be **GPU-resident** (torch tensors or :class:`pgsd.fl.DeviceField`), in which case all of them
are packed by one fused HIP launch and streamed to the file; the per-rank row counts come
from one allgather over the installed communicator (RCCL over xGMI on GPU ranks).
"""
from collections import OrderedDict
import json
import logging
import hashlib
import warnings

import numpy

try:
    from . import fl
except ImportError:  # pragma: no cover - the shared library is missing
    fl = None

from .version import __version__ as _pgsd_version

logger = logging.getLogger('pgsd.hoomd')


_HOST_TYPES = (numpy.ndarray, numpy.generic, int, float, list, tuple, str)


def _is_device(x):
    if isinstance(x, _HOST_TYPES):      # the common case first: no attribute lookups on the torch side
        return False
    return fl is not None and (isinstance(x, fl.DeviceField) or fl._is_device_array(x))


_READER_FIELDS = {}


def _reader_fields(path, container):
    """(name, default, chunk name, upstream-only attribute?) of the per-row chunks of a container, built once."""
    fields = _READER_FIELDS.get(path)
    if fields is None:
        items = [(n, d, False) for n, d in container._default_value.items()]
        if path == 'particles':
            items += [(n, d, True) for n, d in container._extra_default_value.items()]
        fields = tuple((n, d, path + '/' + n, opt) for n, d, opt in items if n not in ('N', 'types', 'type_shapes'))
        _READER_FIELDS[path] = fields
    return fields


_SCHEMA_NAMES = []


def _schema_names(frame):
    """(path, chunk names in the order of the `_default_value` tables) of the three containers, built once."""
    if not _SCHEMA_NAMES:
        for path in ('configuration', 'particles', 'constraints'):
            container = getattr(frame, path)
            names = list(container._default_value)
            if path == 'particles':
                names += list(container._extra_default_value)
            _SCHEMA_NAMES.append((path, tuple(names)))
    return _SCHEMA_NAMES


def _rows(x):
    if fl is not None and isinstance(x, fl.DeviceField):
        return x.N
    shape = getattr(x, 'shape', None)
    if shape is None:                   # a bare __cuda_array_interface__ object
        shape = x.__cuda_array_interface__['shape']
    return int(shape[0])


class ConfigurationData(object):
    """Store configuration data (hoomd.py:45-108).

    Attributes:
        step (int): time step of this frame (:chunk:`configuration/step`).
        dimensions (int): number of dimensions (:chunk:`configuration/dimensions`); defaults to
            2 when the box has Lz == 0, else 3.
        box ((6,) float32): [lx, ly, lz, xy, xz, yz] (:chunk:`configuration/box`).
    """

    _default_value = OrderedDict()
    _default_value['step'] = numpy.uint64(0)
    _default_value['dimensions'] = numpy.uint8(3)
    _default_value['box'] = numpy.array([1, 1, 1, 0, 0, 0], dtype=numpy.float32)

    def __init__(self):
        self.step = None
        self.dimensions = None
        self._box = None

    def _get_box(self):
        return self._box

    def _set_box(self, value):
        # setting a box also fixes `dimensions` while the user has not chosen one: 2 for a flat box
        # (Lz == 0), else 3 (hoomd.py:89-99)
        self._box = value
        if self.dimensions is not None:
            return
        try:
            flat = value[2] == 0
        except TypeError:
            return
        self.dimensions = 2 if flat else 3

    box = property(_get_box, _set_box)

    def validate(self):
        """Convert ``box`` to a (6,) float32 array; ignore attributes that are ``None``."""
        logger.debug('Validating ConfigurationData')
        if self.box is not None:
            self.box = numpy.ascontiguousarray(self.box, dtype=numpy.float32)
            self.box = self.box.reshape([6])


# (dtype, columns) of every per-particle chunk; the first block is the PGSD-SPH schema the
# reference implements (hoomd.py:167-184), the second the upstream HOOMD attributes that the
# reference documents (hoomd.py:133-158) but leaves out of its reader.
_PARTICLE_SPEC = OrderedDict([
    ('typeid', (numpy.uint32, 1)), ('mass', (numpy.float32, 1)), ('body', (numpy.int32, 1)),
    ('position', (numpy.float32, 3)), ('velocity', (numpy.float32, 3)),
    ('slength', (numpy.float32, 1)), ('density', (numpy.float32, 1)), ('pressure', (numpy.float32, 1)),
    ('energy', (numpy.float32, 1)),
    ('auxiliary1', (numpy.float32, 3)), ('auxiliary2', (numpy.float32, 3)),
    ('auxiliary3', (numpy.float32, 3)), ('auxiliary4', (numpy.float32, 3)),
    ('image', (numpy.int32, 3)),
])
_PARTICLE_SPEC_EXTRA = OrderedDict([
    ('charge', (numpy.float32, 1)), ('diameter', (numpy.float32, 1)),
    ('moment_inertia', (numpy.float32, 3)), ('orientation', (numpy.float32, 4)),
    ('angmom', (numpy.float32, 4)),
])


class ParticleData(object):
    """Store particle data chunks (hoomd.py:111-270).

    ``N`` is the number of particles **this rank** holds when writing and the global number
    when reading.  Array attributes may be numpy arrays / array-likes, or GPU-resident torch
    tensors / :class:`pgsd.fl.DeviceField` objects (write only).
    """

    _default_value = OrderedDict()
    _default_value['N'] = numpy.uint32(0)
    _default_value['types'] = ['A']
    _default_value['typeid'] = numpy.uint32(0)
    _default_value['mass'] = numpy.float32(1.0)
    _default_value['body'] = numpy.int32(-1)
    _default_value['position'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['velocity'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['slength'] = numpy.float32(1.0)
    _default_value['density'] = numpy.float32(0.0)
    _default_value['pressure'] = numpy.float32(0.0)
    _default_value['energy'] = numpy.float32(0.0)
    _default_value['auxiliary1'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['auxiliary2'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['auxiliary3'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['auxiliary4'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _default_value['image'] = numpy.array([0, 0, 0], dtype=numpy.int32)
    _default_value['type_shapes'] = [{}]

    # upstream HOOMD attributes: written when set, read back when present in the file
    _extra_default_value = OrderedDict()
    _extra_default_value['charge'] = numpy.float32(0.0)
    _extra_default_value['diameter'] = numpy.float32(1.0)
    _extra_default_value['moment_inertia'] = numpy.array([0, 0, 0], dtype=numpy.float32)
    _extra_default_value['orientation'] = numpy.array([1, 0, 0, 0], dtype=numpy.float32)
    _extra_default_value['angmom'] = numpy.array([0, 0, 0, 0], dtype=numpy.float32)

    def __init__(self):
        self.N = 0
        self.types = None
        self.type_shapes = None
        for name in _PARTICLE_SPEC:
            setattr(self, name, None)
        for name in _PARTICLE_SPEC_EXTRA:
            setattr(self, name, None)

    def validate(self):
        """Convert host arrays to contiguous arrays of the schema dtype and shape
        (hoomd.py:206-270); device-resident attributes are only shape-checked."""
        logger.debug('Validating ParticleData')
        for spec in (_PARTICLE_SPEC, _PARTICLE_SPEC_EXTRA):
            for name, (dt, M) in spec.items():
                value = getattr(self, name)
                if value is None:
                    continue
                if _is_device(value):
                    if _rows(value) != self.N:
                        raise ValueError("particles/%s has %d rows, expected N=%d" % (name, _rows(value), self.N))
                    continue
                value = numpy.ascontiguousarray(value, dtype=dt)
                setattr(self, name, value.reshape([self.N]) if M == 1 else value.reshape([self.N, M]))
        if self.types is not None and (not len(set(self.types)) == len(self.types)):
            raise ValueError("Type names must be unique.")


class BondData(object):
    """Topology container of the upstream schema (hoomd.py:273-362): ``M`` particles per group (2 bonds / pairs, 3
    angles, 4 dihedrals / impropers), ``N``, ``types``, ``typeid`` (N,) uint32, ``group`` (N, M) int32.  The reference
    keeps the class but no :class:`Frame` of it carries topology (``Frame.__init__``, hoomd.py:450-456; the sections
    are commented out of its writer, hoomd.py:585-590), and neither does this one: the class is here so that code
    written against the reference imports and validates the same way."""

    def __init__(self, M):
        self.M = M
        self.N = 0
        self.types = None
        self.typeid = None
        self.group = None
        self._default_value = OrderedDict(
            [('N', numpy.uint32(0)), ('types', []), ('typeid', numpy.uint32(0)),
             ('group', numpy.array([0] * M, dtype=numpy.int32))])

    def validate(self):
        logger.debug('Validating BondData')
        for name, dtype, shape in (('typeid', numpy.uint32, [self.N]), ('group', numpy.int32, [self.N, self.M])):
            value = getattr(self, name)
            if value is not None:
                setattr(self, name, numpy.ascontiguousarray(value, dtype=dtype).reshape(shape))
        if self.types is not None and len(set(self.types)) != len(self.types):
            raise ValueError("Type names must be unique.")


class ConstraintData(object):
    """Store constraint data (hoomd.py:365-421): ``N``, ``value`` (N,) float32, ``group`` (N, 2) int32."""

    def __init__(self):
        self.M = 2
        self.N = 0
        self.value = None
        self.group = None
        self._default_value = OrderedDict()
        self._default_value['N'] = numpy.uint32(0)
        self._default_value['value'] = numpy.float32(0)
        self._default_value['group'] = numpy.array([0] * self.M, dtype=numpy.int32)

    def validate(self):
        logger.debug('Validating ConstraintData')
        if self.value is not None:
            self.value = numpy.ascontiguousarray(self.value, dtype=numpy.float32)
            self.value = self.value.reshape([self.N])
        if self.group is not None:
            self.group = numpy.ascontiguousarray(self.group, dtype=numpy.int32)
            self.group = self.group.reshape([self.N, self.M])


class Frame(object):
    """System state at one point in time (hoomd.py:424-467).

    Attributes:
        configuration (`ConfigurationData`), particles (`ParticleData`),
        constraints (`ConstraintData`), state (dict), log (dict of array-likes).
        num_procs (int): number of writer ranks this frame was partitioned for.
        part_dist: optional integer array with every rank's local particle count; when
            ``None``, `HOOMDTrajectory.append` obtains it with one allgather.
    """

    def __init__(self, num_procs=0):
        self.configuration = ConfigurationData()
        self.particles = ParticleData()
        self.constraints = ConstraintData()
        self.state = {}
        self.log = {}
        self.num_procs = num_procs
        self.part_dist = None

    def validate(self):
        """Validate all contained frame data."""
        self.configuration.validate()
        self.particles.validate()
        self.constraints.validate()


class _FrameCursor(object):
    """What ``iter()`` hands out for a trajectory or a subset of one (the role of hoomd.py:471-488):
    it knows how many frames it covers (``len``), yields them in order, and asking it for an iterator
    again starts a fresh pass over the same frames -- so ``list(cursor)`` can be taken more than once."""

    def __init__(self, trajectory, indices):
        self._trajectory = trajectory
        self._indices = indices
        self._at = 0

    def __len__(self):
        return len(self._indices)

    def __iter__(self):
        return _FrameCursor(self._trajectory, self._indices)

    def __next__(self):
        if self._at >= len(self._indices):
            raise StopIteration
        i = self._indices[self._at]
        self._at += 1
        return self._trajectory[i]


class _FrameSubset(object):
    """The frames of a trajectory picked by a sequence of indices: what slicing a
    `HOOMDTrajectory` returns (the role of hoomd.py:491-512).  Supports ``len``, iteration,
    integer indexing and further slicing; frames are read when they are asked for."""

    def __init__(self, trajectory, indices):
        self._trajectory = trajectory
        self._indices = indices

    def __len__(self):
        return len(self._indices)

    def __iter__(self):
        return _FrameCursor(self._trajectory, self._indices)

    def __getitem__(self, key):
        picked = self._indices[key]
        if isinstance(key, slice):
            return _FrameSubset(self._trajectory, picked)
        return self._trajectory[picked]


def _equal(a, b):
    """`numpy.array_equal`, deciding from the first rows where it can: a 10^6-particle array that differs from
    frame 0 differs in its first rows, and comparing all of it costs as much as writing it."""
    if type(a) is not numpy.ndarray or type(b) is not numpy.ndarray:
        if isinstance(a, (int, float, numpy.number)) and isinstance(b, (int, float, numpy.number)):
            return bool(a == b)             # step, dimensions, N: no array is made of two scalars
        a, b = numpy.asarray(a), numpy.asarray(b)
    if a.shape != b.shape:
        return False
    if a.ndim >= 1 and a.shape[0] > 4096 and not (a[:1024] == b[:1024]).all():
        return False
    return bool((a == b).all())


_DEFAULT_AS = {}


def _equiv(a, default):
    """`numpy.array_equiv(a, default)` (default broadcast over the rows), first rows first.  The schema's defaults
    are Python lists and ints: compared as they are, a float32 array is promoted to float64 element by element
    (18 us for 1 024 positions); the default is therefore cast ONCE to the array's type -- when that is exact."""
    a = numpy.asarray(a)
    try:
        key = (a.dtype, default if not isinstance(default, list) else tuple(default))
        d = _DEFAULT_AS.get(key)
    except TypeError:                   # nested / unhashable default: compare the slow way
        key, d = None, None
    if d is None:
        d = numpy.asarray(default)
        if d.dtype.kind in 'biuf' and a.dtype.kind in 'biuf':
            with numpy.errstate(all='ignore'):
                cast = d.astype(a.dtype)
            if cast.shape == d.shape and (cast == d).all():
                d = cast
        if key is not None:
            _DEFAULT_AS[key] = d
    if d.ndim and a.shape[a.ndim - d.ndim:] != d.shape:
        return bool(numpy.array_equiv(a, d))        # not the scalar / one-row defaults of the schema: numpy decides
    # the first row first: an attribute that is set differs from its default there (and a row-broadcast compare of
    # a whole (N, 3) array costs 3-4x a dense one)
    if a.ndim > d.ndim and a.shape[0] > 1 and not (a[0] == d).all():
        return False
    return bool((a == d).all())


def _encode_strings(strings):
    """list[str] -> (n, wid) int8 array, NUL padded (hoomd.py:628-630)."""
    wid = max(len(w.encode('utf-8')) for w in strings) + 1
    b = numpy.array([w.encode('utf-8') for w in strings], dtype=numpy.dtype((bytes, wid)))
    return b.view(dtype=numpy.int8).reshape(len(b), wid)


def _shape_key(name, value):
    """(type, shape) of a replicated / state / log value as it will be written: what must agree between the ranks."""
    if isinstance(value, (list, tuple)) and name in ('types', 'type_shapes'):
        return ('strings', len(value), max([len(str(v).encode('utf-8')) for v in value] or [0]))
    a = numpy.asarray(value)
    return (a.dtype.str, a.shape)


class _AppendPlan(object):
    """One frame on its way through `HOOMDTrajectory.append`: `_plan_frame` fills it, `_agree` settles it over the
    ranks, `_write_frame` carries it out."""
    __slots__ = ('entries', 'dev', 'host_pp', 'replicated', 'at_count', 'n_local', 'part0', 'ticket', 'compared',
                 'frame0_equal', 'part_dist', 'n_global', 'declared')

    def __init__(self):
        self.entries = []       # [path, name, write?] per schema name, in the reference's chunk order
        self.dev = []           # GPU-resident per-particle attributes: (position in entries, chunk name, DeviceField)
        self.host_pp = []       # several ranks: per-particle HOST arrays that would be written: (position, chunk, array)
        self.replicated = []    # (chunk name, shape key) of the replicated values that are set on this rank
        self.at_count = None    # position of particles/N
        self.n_local = 0
        self.part0 = None       # the partition this rank's rows of frame 0 belong to (None: no such comparisons now)
        self.ticket = None      # staging ticket of the device arrays (None: they leave in fused launches)
        self.compared = []      # positions in `dev` that were compared with this rank's rows of frame 0 on the GPU
        self.frame0_equal = []  # positions in `entries` elided because they equal THIS rank's rows of frame 0
        self.part_dist = None
        self.n_global = 0
        self.declared = False


class HOOMDTrajectory(object):
    """Read and write hoomd pgsd files (hoomd.py:515-940).

    Args:
        file (`pgsd.fl.PGSDFile` or `pgsd.pypgsd.PGSDFile`): file to access.

    Attributes:
        device_elision: `append` applies the elision rule of hoomd.py:654-694 to GPU-resident per-particle arrays ON
            THE GPU.  True (default; ``'exact'`` is the same): every array that is set is compared in every frame --
            with this rank's rows of frame 0 where frame 0 holds the chunk, with the default value where it does
            not -- and the file is the one host arrays of the same values give, for every input (NaNs, signed zeros,
            arrays that return to frame 0's values, default-valued arrays); the price is one frame's worth of HBM for
            the rows of frame 0 and one comparison launch per frame.  False: GPU-resident arrays are always written.
            (Round 3's ``'once'`` shortcut -- stop comparing an array once it differed -- was removed in round 5: it
            wrote arrays that return to frame 0's values where the host path elides them.)

    Per-particle attributes of a `Frame` may be numpy arrays, torch GPU tensors or `pgsd.fl.DeviceField` views of
    GPU memory (a column range of a ``Scalar4`` array, a converted or bit-cast element type); `append` writes the
    GPU-resident ones through the fused pack kernel, `read_frame_device` restores a partition into GPU arrays.
    """

    def __init__(self, file):
        if file.mode == 'ab':
            raise ValueError('Append mode not yet supported')
        self._file = file
        self._initial_frame = None
        self._elision_ref = None       # what append compares against (frame 0; several ranks: its replicated part)
        self._elision_lazy = False     # one rank: per-particle arrays of frame 0 are read when first compared
        self._frame0_chunks = None
        #: compare GPU-resident per-particle arrays with frame 0 on the GPU and elide the equal ones, as hoomd.py:654-694
        #: does for host arrays (`append`); False: GPU-resident arrays are always written
        self.device_elision = True
        self._dev_ref = {}             # chunk -> GPU tensor: this rank's rows of frame 0 as the chunk stores them
        self._dev_ref_part = None      # the partition (every rank's row count) those rows belong to
        self._default_ref = {}         # (chunk, device) -> GPU tensor: rows of the default value as the chunk stores them
        self._dev_off = False          # the partition changed: frame 0's rows are other particles' from now on
        self._host_ref = {}            # several ranks: chunk -> this rank's rows of frame 0 (host arrays are compared too)
        self._prefix_names = {}        # reader: prefix -> (nnames when asked, matching chunk names)
        self._frame0_small_cache = {}  # device reader: small replicated chunks of frame 0, read once
        self._frame0_dev_part = None   # read_frame_device: the partition whose rows of frame 0 are kept in HBM ...
        self._frame0_dev_cache = {}    # ... chunk -> GPU tensor
        logger.info('opening HOOMDTrajectory: ' + str(self.file))
        if self.file.schema != 'hoomd':
            raise RuntimeError('PGSD file is not a hoomd schema file: ' + str(self.file))
        version = self.file.schema_version
        if not (version < (2, 0) and version >= (1, 0)):
            raise RuntimeError('Incompatible hoomd schema version ' + str(version) + ' in: ' + str(self.file))
        logger.info('found ' + str(len(self)) + ' frames')

    @property
    def file(self):
        """The file handle."""
        return self._file

    def __len__(self):
        """The number of frames in the trajectory."""
        return self.file.nframes

    # ------------------------------------------------------------------ writing
    def _comm(self):
        # the communicator the FILE was opened on (the process default unless `fl.open(..., comm=)` chose another)
        f = self.file
        if hasattr(f, 'nprocs'):
            return f.rank, f.nprocs
        return 0, 1                     # pgsd.pypgsd.PGSDFile: the pure-Python reader, one rank

    def append(self, frame, wait=True):
        """Append a frame (collective over the ranks of the file's communicator).

        ``wait=False`` seals the frame asynchronously (`pgsd.fl.PGSDFile.end_frame`): call
        ``trajectory.file.wait_packed()`` before changing GPU-resident arrays of the frame and
        ``trajectory.file.frame_sync()`` before relying on the file.

        The calls this makes are the ones the reference sketches (hoomd.py:569-642), pinned byte for
        byte against the oracle by ``tests/test_hoomd_append_oracle.py``:

        * order: ``configuration/{step, dimensions, box}``, then ``particles/*`` and ``constraints/*``
          in the order of their ``_default_value`` tables (hoomd.py:60-63, 167-184), then ``state/*``,
          then ``log/*``, then ``end_frame``;
        * ``box``, ``N`` (uint32, ``part_dist.sum()``), ``step`` (uint64), ``dimensions`` (uint8),
          ``types`` and ``type_shapes`` (int8 ``n x wid``, NUL padded, JSON for the shapes) are
          replicated small chunks: ``write_all=False, offset=None`` (hoomd.py:604-630);
        * per-particle arrays: ``write_all=True, offset=part_dist`` (hoomd.py:597-600);
        * ``state/*`` and ``log/*``: ``write_chunk(name, data)`` with default arguments (hoomd.py:634-640);
        * fields that are ``None`` are not written; fields that equal the initial frame, or the default value where
          frame 0 has no such chunk, are elided as hoomd.py:654-694 describes -- host arrays and GPU-resident arrays
          alike, by the same rule (``numpy.array_equal``'s equality: a NaN equals nothing, +0.0 equals -0.0).

        Three steps (`_plan_frame`, `_agree`, `_write_frame`): every rank decides what it would write, ONE allgather
        carries the ranks' row counts and votes, the chunks are written in the reference's order.

        Where the sketch cannot be followed literally:

        * ``constraints/*`` are replicated small chunks and ``constraints/N`` is the constraint count
          (the sketch would write the PARTICLE count as ``constraints/N`` and partition ``value`` /
          ``group`` by the particle distribution, hoomd.py:597-611: files its own reader rejects);
        * the write/skip decision of every chunk is agreed over the ranks (written if any rank needs
          it; a rank without a value contributes the default for its rows), because a chunk write is
          collective; ``particles/N`` is compared as the global count;
        * with several ranks a per-particle array is compared with THIS rank's rows of frame 0 (against the whole of
          frame 0, as the sketch compares, it could never be equal), while the partition is the one those rows were
          written with; after a change of the partition every per-particle array that is set is written;
        * GPU-resident per-particle arrays are compared ON THE GPU (`device_elision`): packed by one launch, their
          packed rows compared with this rank's rows of frame 0 (or with rows of the default value) in device memory;
        * replicated chunks, ``state/*`` and ``log/*`` must have the same shape on every rank (checked in the vote);
        * upstream HOOMD attributes (charge, diameter, moment_inertia, orientation, angmom) follow the
          SPH set in that order when they are set.
        """
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug('Appending frame to hoomd trajectory: ' + str(self.file))
        frame.validate()
        rank, size = self._comm()
        self._elision_reference(frame, size)
        # From here on the frame costs ONE collective however many chunks of whatever kind it has: the allgather of
        # `_agree` (every rank's row count + its write/skip votes).  With the counts in hand the partition is DECLARED
        # to the library (pgsd_set_partition), which then places every chunk without an exchange of its own:
        # per-particle chunks by the declared rows (offset='auto'), everything else -- replicated small chunks,
        # state/* and log/* -- has the same size on every rank.  The file is byte-identical to the one the
        # exchanges would produce.  (One rank: nothing to exchange, the batched queue keeps the call pattern.)
        if size == 1 and getattr(self.file, 'frame_exchange', None) is False:
            self.file.frame_exchange = True
        plan = self._plan_frame(frame, rank, size)
        self._agree(plan, frame, rank, size)
        self._write_frame(plan, frame, rank, size, wait)

    # -- step 0 (once per trajectory): what the elision rule compares with
    def _elision_reference(self, frame, size):
        """The initial frame is the reference for elision (hoomd.py:654-694).  With several ranks only its REPLICATED
        part is read: a per-particle array of frame 0 holds all ranks' rows and can never equal this rank's share
        (round 2 had every rank read the whole global frame here -- 2.24 GB each at 8 x 10 M particles -- to find
        that out).  One rank: a per-particle array of frame 0 is read when (and if) a host array is to be compared with
        it -- reading all of frame 0 here cost the second append of a 10 M-particle trajectory 250 ms, for arrays that
        live on the GPU and are compared there never to be looked at.

        ... and so is the set of chunks frame 0 holds (hoomd.py:689-691).  Looked up ONCE, by every rank, for every
        name in the same order -- never from inside a comparison only some ranks make: a lookup flushes whatever is
        pending, which is collective (a rank whose velocities are all zero would ask alone)."""
        if len(self) == 0:
            return
        if self._elision_ref is None:
            if size == 1 and self._initial_frame is not None:
                self._elision_ref = self._initial_frame
            else:
                self._elision_ref = self._read_frame0_replicated()
                self._elision_lazy = size == 1
        if self._frame0_chunks is None:
            self._frame0_chunks = set()
            for path, names in _schema_names(frame):
                for name in names:
                    if self.file.chunk_exists(frame=0, name=path + '/' + name, write_all=False):
                        self._frame0_chunks.add(path + '/' + name)

    # -- step 1: this rank's decisions
    def _plan_frame(self, frame, rank, size):
        """What this rank would write: one entry per schema name, in the reference's order."""
        plan = _AppendPlan()
        entries = plan.entries
        dev, host_pp = plan.dev, plan.host_pp
        for path, names in _schema_names(frame):
            values = getattr(frame, path).__dict__
            particles = path == 'particles'
            for name in names:
                if particles and name == 'N':
                    # None = "as in frame 0": no count chunk; decided in `_agree` from the global count otherwise
                    plan.at_count = len(entries)
                    entries.append([path, name, frame.particles.N is not None])
                    continue
                value = values.get(name)
                if value is None and name == 'box':
                    value = values.get('_box')              # (the one attribute kept behind a property)
                if value is None:
                    entries.append([path, name, False])     # most of the schema, most of the time: not set
                    continue
                per_particle = particles and (name in _PARTICLE_SPEC or name in _PARTICLE_SPEC_EXTRA)
                if per_particle and _is_device(value):
                    dt, _ = (_PARTICLE_SPEC.get(name) or _PARTICLE_SPEC_EXTRA.get(name))
                    field = value if isinstance(value, fl.DeviceField) else fl.DeviceField.from_device_array(value, out_dtype=dt)
                    dev.append((len(entries), path + '/' + name, field))
                    entries.append([path, name, True])
                    continue
                write = self._should_write(path, name, frame, None)
                if per_particle:
                    if write and size > 1:
                        host_pp.append((len(entries), path + '/' + name, value))
                else:
                    plan.replicated.append((path + '/' + name, _shape_key(name, value)))
                entries.append([path, name, write])
        plan.n_local = int(frame.particles.N) if frame.particles.N is not None else 0
        plan.part0 = self._frame0_partition(frame, dev or host_pp, rank, size, plan.n_local)
        self._host_row_votes(plan, rank)
        self._device_votes(plan, rank)
        return plan

    # -- step 2: the frame's ONE collective
    def _agree(self, plan, frame, rank, size):
        """One allgather carries every rank's particle count (-> part_dist, the MPI_Allgather of
        benchmark-write.cc:41), a digest of the shapes of its replicated / state / log chunks, and two bits per plan
        entry: "write" and "elided only because it equals THIS rank's rows of frame 0".  Everything decided from here
        on rests on replicated data, so every rank decides alike."""
        entries = plan.entries
        n_local = plan.n_local
        if frame.part_dist is not None:
            part_dist = numpy.asarray(frame.part_dist, dtype=numpy.uint64)
            if part_dist.shape[0] != size:
                raise ValueError("part_dist must have one entry per rank")
        else:
            part_dist = numpy.array([n_local], dtype=numpy.uint64)
        rows_equal = numpy.zeros(len(entries), dtype=bool)      # any rank: elided against its rows of frame 0
        for at in plan.frame0_equal:
            rows_equal[at] = True
        if size > 1:
            mine = numpy.zeros(16 + len(entries), dtype=numpy.uint8)
            mine[:8] = numpy.array([n_local], dtype=numpy.uint64).view(numpy.uint8)
            digest = hashlib.blake2b(repr(plan.replicated + [('state/' + k, _shape_key(k, v)) for k, v in frame.state.items()]
                                          + [('log/' + k, _shape_key(k, v)) for k, v in frame.log.items()]).encode(),
                                     digest_size=8).digest()
            mine[8:16] = numpy.frombuffer(digest, dtype=numpy.uint8)
            mine[16:] = [(1 if e[2] else 0) | (2 if rows_equal[at] else 0) for at, e in enumerate(entries)]
            allb = self.file.allgather(mine)
            if frame.part_dist is None:
                part_dist = numpy.ascontiguousarray(allb[:, :8]).view(numpy.uint64).reshape(size)
            if (allb[:, 8:16] != allb[0, 8:16]).any():
                # placed without an exchange (declared partition), such chunks would put different offsets into each
                # rank's replicated index: every rank sees the same vector, every rank stops here
                raise ValueError("replicated chunks (configuration/*, types, constraints/*, state/*, log/*) must be set "
                                 "on every rank with the same shape and type: the ranks differ")
            bits = allb[:, 16:]
            for e, w in zip(entries, (bits & 1).max(axis=0)):
                e[2] = bool(w)
            rows_equal = (bits & 2).max(axis=0).astype(bool)
        plan.part_dist = part_dist
        plan.n_global = int(part_dist.sum())
        if plan.at_count is not None and entries[plan.at_count][2]:
            entries[plan.at_count][2] = self._should_write('particles', 'N', frame, plan.n_global)
        self._partition_bookkeeping(plan, rows_equal, size)
        plan.declared = size > 1 and hasattr(self.file, 'set_partition')
        if plan.declared and int(part_dist[rank]) != n_local:
            raise ValueError("part_dist[%d] = %d but this rank holds %d particles" % (rank, int(part_dist[rank]), n_local))

    def _partition_bookkeeping(self, plan, rows_equal, size):
        """Comparisons with this rank's rows of frame 0 hold while the partition is the one those rows were written
        with.  Decided from the gathered partition and the gathered "equal to my rows of frame 0" bits alone --
        replicated data (the override used to hang on a rank-local condition, so that one rank placed a
        chunk the others did not)."""
        part = tuple(int(x) for x in plan.part_dist)
        n0 = self._elision_ref.particles.N if self._elision_ref is not None else None
        fits = len(self) == 0 or (n0 is not None and int(n0) == sum(part))     # frame 0 has that many particles
        if size == 1:
            # one rank: "my rows of frame 0" are the whole of frame 0 -- comparable whenever the counts agree, in
            # whatever frame (`_frame0_partition`), exactly as host arrays are compared (`_should_write`)
            if self._dev_ref_part is None and fits:
                self._dev_ref_part = part
        elif not self._dev_off:
            if self._dev_ref_part is None:
                # frame 0 being written, or the first frame appended to an existing file: the rows of frame 0 kept /
                # read from now on are those of THIS partition
                if fits:
                    self._dev_ref_part = part
                else:
                    self._dev_off = True
            elif part != self._dev_ref_part:
                # particles moved between the ranks (or their number changed): whatever was compared was compared
                # with other particles' rows.  Every per-particle array that some rank elided on those grounds is
                # written, and every array that is set from now on
                self._dev_off = True
                for at, e in enumerate(plan.entries):
                    if rows_equal[at]:
                        e[2] = True
        if self._dev_off:
            self._dev_ref.clear()
            self._host_ref.clear()
            return

    # -- step 3: the chunks, in the reference's order
    def _write_frame(self, plan, frame, rank, size, wait):
        debug = logger.isEnabledFor(logging.DEBUG)
        part_dist, n_local, n_global = plan.part_dist, plan.n_local, plan.n_global
        if plan.declared:
            self.file.set_partition(part_dist)
        particle_offset = 'auto' if plan.declared else part_dist
        ticket = plan.ticket
        if ticket is not None and len(self) == 0 and self._dev_ref_part is not None:
            # frame 0: the packed rows of what is written stay in HBM for the comparisons to come (a device-to-device
            # copy behind the pack)
            sizes = [int(f.N) * int(f.M) * f.out_dtype.itemsize if plan.entries[at][2] else None for at, _, f in plan.dev]
            for (_, chunk, _), ref in zip(plan.dev, self.file.copy_staged(ticket, 0, sizes)):
                if ref is not None:
                    self._dev_ref[chunk] = ref
        device_fields = []      # not staged: consecutive GPU-resident fields leave in one fused launch
        staged_run = []         # staged (one launch for the whole frame): consecutive chunk numbers of the ticket
        dev_at = dict((at, (k, field)) for k, (at, _, field) in enumerate(plan.dev)) if plan.dev else {}
        for at, (path, name, write) in enumerate(plan.entries):
            if not write:
                continue
            container = getattr(frame, path)
            data = getattr(container, name)
            chunk = path + '/' + name
            if debug:
                logger.debug('writing data chunk: ' + chunk)
            if path == 'particles' and (name in _PARTICLE_SPEC or name in _PARTICLE_SPEC_EXTRA):
                dt, M = (_PARTICLE_SPEC.get(name) or _PARTICLE_SPEC_EXTRA.get(name))
                if at in dev_at:
                    k, field = dev_at[at]
                    if ticket is None:
                        device_fields.append((chunk, field))
                    else:
                        if staged_run and staged_run[-1] + 1 != k:
                            self._flush_staged(ticket, staged_run, particle_offset, rank)
                        staged_run.append(k)
                    continue
                if data is None:
                    # another rank needs the chunk: contribute this rank's rows of the default
                    default = container._default_value.get(name, container._extra_default_value.get(name))
                    data = numpy.empty([n_local] + ([M] if M > 1 else []), dtype=dt)
                    data[...] = default
                self._flush_device_fields(device_fields, particle_offset, rank)
                self._flush_staged(ticket, staged_run, particle_offset, rank)
                self.file.write_chunk(chunk, data, particle_offset, rank, True)
                continue
            self._flush_device_fields(device_fields, particle_offset, rank)
            self._flush_staged(ticket, staged_run, particle_offset, rank)
            # replicated small chunks (hoomd.py:604-630)
            if name == 'N':
                count = n_global if path == 'particles' else int(container.N)
                data = numpy.array([count], dtype=numpy.uint32)
            elif name == 'step':
                data = numpy.array([data], dtype=numpy.uint64)
            elif name == 'dimensions':
                data = numpy.array([data], dtype=numpy.uint8)
            elif name in ('types', 'type_shapes'):
                if name == 'type_shapes':
                    data = [json.dumps(shape_dict) for shape_dict in data]
                data = _encode_strings(data)
            self.file.write_chunk(chunk, data, None, rank, False)
        self._flush_device_fields(device_fields, particle_offset, rank)
        self._flush_staged(ticket, staged_run, particle_offset, rank)

        # state and logged quantities: the sketch calls ``write_chunk(name, data)`` with the binding's
        # default arguments (hoomd.py:634-640; the state loop is commented out twice, upstream GSD
        # has it) -- write_all=True, no offset: every rank writes its (replicated) value at the same
        # place and the file advances by the ranks' sizes summed (pgsd.c:2240-2246)
        for state, data in frame.state.items():
            self.file.write_chunk('state/' + state, numpy.ascontiguousarray(data))
        for log, data in frame.log.items():
            self.file.write_chunk('log/' + log, data)

        self.file.end_frame(wait=wait)
        if plan.declared:
            self.file.set_partition(None)       # the declaration was this frame's

    def _flush_staged(self, ticket, run, part_dist, rank):
        if run:
            self.file.write_staged(ticket, run[0], len(run), offset=part_dist, rank=rank)
            del run[:]

    def _frame0_partition(self, frame, wanted, rank, size, n_local):
        """The partition (every rank's row count) for which this rank's rows of frame 0 are, or can be, at hand -- or
        None when per-particle arrays cannot be compared with frame 0's rows in this frame: nothing to compare, frame 0
        itself, the comparisons ended (`_dev_off`), a partition that differs from the one the rows were taken for,
        or one that is not known before the frame's exchange (no `Frame.part_dist`, several ranks, first frame
        appended to an existing file: the comparisons then start with the next frame)."""
        if not wanted or self._dev_off or len(self) == 0:
            return None
        given = None
        if frame.part_dist is not None:
            given = tuple(int(x) for x in numpy.asarray(frame.part_dist).reshape(-1))
        elif size == 1:
            given = (n_local,)
        part = self._dev_ref_part
        if part is None:
            n0 = self._elision_ref.particles.N if self._elision_ref is not None else None
            if given is None or n0 is None or int(n0) != sum(given):
                return None
            part = given
        elif given is not None and given != part:
            return None                     # (`_partition_bookkeeping` sees the change and ends the comparisons)
        if len(part) != size or part[rank] != n_local:
            return None
        return part

    def _read_frame0_rows(self, chunk, row0, n, device):
        """This rank's rows ``[row0, row0 + n)`` of a per-particle chunk of frame 0, as a numpy array or a GPU tensor.
        A LOCAL read (`PGSDFile.local_reads`, lookup included): these are rows this rank wrote itself in this session,
        or rows of a file that was complete when it was opened -- and which arrays a rank compares need not be the
        same on every rank, so the read must not be a collective."""
        f = self.file
        before = f.local_reads
        f.local_reads = True
        try:
            if device:
                return f.read_chunk_device(0, chunk, N=n, offset=row0)
            return f.read_rows(0, chunk, row0, n)
        finally:
            f.local_reads = before

    def _host_row_votes(self, plan, rank):
        """Several ranks: a per-particle HOST array is compared with THIS RANK'S rows of frame 0 (read from the file
        when first needed, ``numpy.array_equal`` as on one rank); equal rows vote "skip".  (Compared with the whole of
        frame 0 -- every rank's rows -- as the sketch has it, hoomd.py:673-687, an array could never be equal.)"""
        if not plan.host_pp or plan.part0 is None:
            return
        frame0 = self._frame0_chunks or ()
        row0 = sum(plan.part0[:rank])
        for at, chunk, data in plan.host_pp:
            if chunk not in frame0:
                continue
            ref = self._host_ref.get(chunk)
            if ref is None:
                ref = self._read_frame0_rows(chunk, row0, plan.n_local, False)
                self._host_ref[chunk] = ref
            if _equal(ref, data):
                logger.debug('skipping data chunk, this rank\'s rows match frame 0: ' + chunk)
                plan.entries[at][2] = False
                plan.frame0_equal.append(at)

    def _default_rows(self, chunk, field, device):
        """Rows of the schema's default value for ``chunk`` as the chunk stores them, in device memory: what "equals
        the default" is tested against (hoomd.py:692-693).  4096 rows stand for any number (the comparison lets a
        short reference repeat); fewer rows than that get exactly as many."""
        key = (chunk, device)
        ref = self._default_ref.get(key)
        if ref is None:
            name = chunk.split('/', 1)[1]
            default = ParticleData._default_value.get(name, ParticleData._extra_default_value.get(name))
            row = numpy.empty((1, int(field.M)), dtype=field.out_dtype)
            row[...] = default
            # 4096 rows of the default value in the library's own device memory: ONE row repeated by the allocation
            ref = fl.DeviceBuffer((4096 * row.nbytes,), numpy.uint8, device, pattern=row)
            self._default_ref[key] = ref
        want = int(field.N) * int(field.M) * field.out_dtype.itemsize
        return ref if want >= ref.nbytes else ref.view(shape=(want,))

    def _device_votes(self, plan, rank):
        """The elision rule of hoomd.py:654-694 for GPU-resident per-particle arrays, decided on the GPU exactly as
        `_should_write` decides it for host arrays: a chunk frame 0 holds is compared with this rank's rows of frame 0
        (equal: skip), a chunk frame 0 does not hold -- or frame 0 itself -- with rows of the default value (equal:
        skip).  Equality is ``numpy.array_equal``'s (`PGSDFile.compare_staged`): the file is the one host arrays of
        the same values give.

        All of the frame's GPU-resident arrays are packed by ONE launch into staging (`stage_chunks`) and compared by
        one more (`compare_staged`: one stream wait).  Frame 0's rows come from the staging of frame 0 itself when
        this trajectory wrote it (`copy_staged` in `_write_frame`: a device-to-device copy), from the file otherwise
        (`read_chunk_device`, once per array).  ``device_elision = False``: no comparisons, every array that is set
        is written.  Everything the comparison keeps in HBM is the library's own memory (`pgsd.fl.DeviceBuffer`) on
        the GPU the file's pipeline runs on: no tensor library is involved, whatever kind of array the sources are."""
        f = self.file
        dev = plan.dev
        if not dev or not self.device_elision or not hasattr(f, 'compare_staged'):
            return
        frame0 = (self._frame0_chunks or ()) if len(self) > 0 else ()
        part0 = plan.part0
        refs = [None] * len(dev)
        kinds = [None] * len(dev)                   # 'rows': against frame 0's rows; 'default': against the default
        device = None
        for k, (_, chunk, field) in enumerate(dev):
            if chunk in frame0:
                if part0 is not None and not self._dev_off:
                    kinds[k] = 'rows'
            else:
                kinds[k] = 'default'
        if not any(kinds):
            return
        ticket = f.stage_chunks([(chunk, field) for _, chunk, field in dev])
        device = ticket[3]              # the GPU the file's pipeline runs on (the handle's own answer)
        row0 = sum(part0[:rank]) if part0 is not None else 0
        for k, (_, chunk, field) in enumerate(dev):
            if kinds[k] == 'default':
                refs[k] = self._default_rows(chunk, field, device)
            elif kinds[k] == 'rows':
                ref = self._dev_ref.get(chunk)
                if ref is None:
                    ref = self._read_frame0_rows(chunk, row0, plan.n_local, True)
                    self._dev_ref[chunk] = ref
                if ref.numel() * ref.element_size() == int(field.N) * int(field.M) * field.out_dtype.itemsize:
                    refs[k] = ref
        equal = f.compare_staged(ticket, 0, refs)
        for k, (at, chunk, _) in enumerate(dev):
            if refs[k] is None:
                continue
            if kinds[k] == 'rows':
                plan.compared.append(k)
            if equal[k]:
                plan.entries[at][2] = False
                if kinds[k] == 'rows':
                    plan.frame0_equal.append(at)
        plan.ticket = ticket

    def _flush_device_fields(self, device_fields, part_dist, rank):
        if device_fields:
            self.file.write_chunks(list(device_fields), offset=part_dist, rank=rank)
            del device_fields[:]

    def _should_write(self, path, name, frame, n_global):
        """False if the value is None, matches the initial frame, or matches the default and
        frame 0 has no such chunk (hoomd.py:654-694)."""
        container = getattr(frame, path)
        data = getattr(container, name, None)
        if name == 'N' and path == 'particles':
            data = n_global
        if data is None:
            return False
        if _is_device(data):
            return True
        if self._elision_ref is not None:
            initial_container = getattr(self._elision_ref, path)
            initial_data = getattr(initial_container, name, None)
            if (initial_data is None and self._elision_lazy and path == 'particles'
                    and (name in _PARTICLE_SPEC or name in _PARTICLE_SPEC_EXTRA)
                    and (path + '/' + name) in (self._frame0_chunks or ())):
                initial_data = self.file.read_chunk(frame=0, name=path + '/' + name, offset=numpy.uint32(0), r_all=False)
                initial_container.__dict__[name] = initial_data
            if initial_data is not None and _equal(initial_data, data):
                logger.debug('skipping data chunk, matches frame 0: ' + path + '/' + name)
                return False
        # a value that matches the default is elided -- unless frame 0 holds the chunk (a reader would take frame 0's
        # value for the default).  Most chunks that are written at all are in frame 0: the lookup comes first, the
        # comparison with the default (the dearer of the two) only when it can decide something
        if (path + '/' + name) in (self._frame0_chunks or ()):
            return True
        default = container._default_value.get(name)
        if default is None and path == 'particles':
            default = container._extra_default_value.get(name)
        if name in ('types', 'type_shapes'):
            matches_default_value = data == default
        else:
            matches_default_value = _equiv(data, default)
        if matches_default_value:
            logger.debug('skipping data chunk, default value: ' + path + '/' + name)
            return False
        return True

    def _read_frame0_replicated(self):
        """Frame 0's replicated values -- configuration, particle count, types, type shapes, constraints -- for the
        elision test of `append` on several ranks; per-particle attributes stay None (never compared)."""
        f = self.file
        snap = Frame()
        self._read_scalar_any(0, 'configuration/step', snap.configuration, 'step')
        self._read_scalar_any(0, 'configuration/dimensions', snap.configuration, 'dimensions')
        snap.configuration.box = f.read_chunk(0, 'configuration/box') if f.chunk_exists(0, 'configuration/box') \
            else snap.configuration._default_value['box']
        for path in ('particles', 'constraints'):
            container = getattr(snap, path)
            container.N = int(f.read_chunk(0, path + '/N')[0]) if f.chunk_exists(0, path + '/N') else 0
        for name in ('types', 'type_shapes'):
            if f.chunk_exists(0, 'particles/' + name):
                tmp = f.read_chunk(0, 'particles/' + name)
                tmp = tmp.view(dtype=numpy.dtype((bytes, tmp.shape[1]))).reshape([tmp.shape[0]])
                if name == 'types':
                    snap.particles.types = list(a.decode('UTF-8') for a in tmp)
                else:
                    snap.particles.type_shapes = list(json.loads(a.decode('UTF-8')) for a in tmp)
            else:
                setattr(snap.particles, name, snap.particles._default_value[name])
        for name in ('value', 'group'):
            if f.chunk_exists(0, 'constraints/' + name):
                setattr(snap.constraints, name, f.read_chunk(0, 'constraints/' + name))
        return snap

    def extend(self, iterable):
        """Append each item of the iterable to the file."""
        for item in iterable:
            self.append(item)

    def close(self):
        """Close the file."""
        self.file.close()
        del self._initial_frame
        self._elision_ref = None
        self._dev_ref = {}
        self._host_ref = {}
        self._frame0_dev_cache = {}

    def flush(self):
        """Flush all buffered frames to the file."""
        self._file.flush()

    # ------------------------------------------------------------------ reading
    def read_frame(self, idx):
        """Read the frame at the given index (deprecated alias of ``trajectory[idx]``)."""
        warnings.warn("Deprecated, trajectory[idx]", DeprecationWarning)
        return self._read_frame(idx)

    def _names_with_prefix(self, prefix):
        """`find_matching_chunk_names(prefix)`, asked again only when the file's name list has grown (names are never
        removed): the reader looks `log/` and `state/` up for every frame."""
        f = self.file
        n = getattr(f, 'nnames', None)
        if n is None:                       # pgsd.pypgsd.PGSDFile: the pure-Python reader keeps no count
            return f.find_matching_chunk_names(prefix, False)
        cached = self._prefix_names.get(prefix)
        if cached is None or cached[0] != n:
            cached = (n, f.find_matching_chunk_names(prefix, False))
            self._prefix_names[prefix] = cached
        return cached[1]

    def _read_scalar(self, idx, chunk, container, attr, fallback_path):
        if self.file.chunk_exists(frame=idx, name=chunk, write_all=False):
            arr = self.file.read_chunk(frame=idx, name=chunk, offset=numpy.uint32(0), r_all=False)
            setattr(container, attr, arr[0])
        elif self._initial_frame is not None:
            setattr(container, attr, getattr(getattr(self._initial_frame, fallback_path), attr))
        else:
            setattr(container, attr, container._default_value[attr])

    def _read_frame(self, idx):
        """Read one frame; chunks missing in the frame come from frame 0 (when N matches) or
        from the default values, returned read-only (hoomd.py:724-902)."""
        if idx >= len(self):
            raise IndexError
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug('reading frame ' + str(idx) + ' from: ' + str(self.file))
        if self._initial_frame is None and idx != 0:
            self._read_frame(0)

        snap = Frame()
        self._read_scalar(idx, 'configuration/step', snap.configuration, 'step', 'configuration')
        self._read_scalar(idx, 'configuration/dimensions', snap.configuration, 'dimensions', 'configuration')
        if self.file.chunk_exists(frame=idx, name='configuration/box', write_all=False):
            snap.configuration.box = self.file.read_chunk(frame=idx, name='configuration/box',
                                                          offset=numpy.uint32(0), r_all=False)
        elif self._initial_frame is not None:
            snap.configuration.box = self._initial_frame.configuration.box
        else:
            snap.configuration.box = snap.configuration._default_value['box']

        for path in ('particles', 'constraints'):
            container = getattr(snap, path)
            initial = getattr(self._initial_frame, path) if self._initial_frame is not None else None

            container.N = 0
            if self.file.chunk_exists(frame=idx, name=path + '/N', write_all=False):
                container.N = self.file.read_chunk(frame=idx, name=path + '/N', offset=numpy.uint32(0),
                                                   r_all=False)[0]
            elif initial is not None:
                container.N = initial.N

            if 'types' in container._default_value:
                if self.file.chunk_exists(frame=idx, name=path + '/types', write_all=False):
                    tmp = self.file.read_chunk(frame=idx, name=path + '/types', offset=numpy.uint32(0), r_all=False)
                    tmp = tmp.view(dtype=numpy.dtype((bytes, tmp.shape[1]))).reshape([tmp.shape[0]])
                    container.types = list(a.decode('UTF-8') for a in tmp)
                elif initial is not None:
                    container.types = initial.types
                else:
                    container.types = container._default_value['types']

            if 'type_shapes' in container._default_value and path == 'particles':
                if self.file.chunk_exists(frame=idx, name=path + '/type_shapes', write_all=False):
                    tmp = self.file.read_chunk(frame=idx, name=path + '/type_shapes', offset=numpy.uint32(0),
                                               r_all=False)
                    tmp = tmp.view(dtype=numpy.dtype((bytes, tmp.shape[1]))).reshape([tmp.shape[0]])
                    container.type_shapes = list(json.loads(s.decode('UTF-8')) for s in tmp)
                elif initial is not None:
                    container.type_shapes = initial.type_shapes
                else:
                    container.type_shapes = container._default_value['type_shapes']

            for name, default, chunk, is_optional in _reader_fields(path, container):
                if self.file.chunk_exists(frame=idx, name=chunk, write_all=False):
                    container.__dict__[name] = self.file.read_chunk(frame=idx, name=chunk, offset=numpy.uint32(0),
                                                                    r_all=False)
                    continue
                if is_optional and (initial is None or initial.__dict__.get(name) is None):
                    continue  # upstream-only attribute that this file never stored
                if initial is not None and initial.N == container.N and initial.__dict__.get(name) is not None:
                    container.__dict__[name] = initial.__dict__[name]
                else:
                    tmp = numpy.array([default])
                    s = list(tmp.shape)
                    s[0] = container.N
                    container.__dict__[name] = numpy.empty(shape=s, dtype=tmp.dtype)
                    container.__dict__[name][:] = tmp
                container.__dict__[name].flags.writeable = False

        for log in self._names_with_prefix('log/'):
            if self.file.chunk_exists(frame=idx, name=log, write_all=False):
                snap.log[log[4:]] = self.file.read_chunk(frame=idx, name=log, offset=numpy.uint32(0), r_all=False)
            elif self._initial_frame is not None and log[4:] in self._initial_frame.log:
                snap.log[log[4:]] = self._initial_frame.log[log[4:]]

        # state chunks belong to the frame they were written in (no fall-back, as upstream GSD reads them)
        for state in self._names_with_prefix('state/'):
            if self.file.chunk_exists(frame=idx, name=state, write_all=False):
                snap.state[state[6:]] = self.file.read_chunk(frame=idx, name=state, offset=numpy.uint32(0), r_all=False)

        if self._initial_frame is None and idx == 0:
            self._initial_frame = snap
        return snap

    def read_frame_device(self, idx, part=None, scalar4=False, defaults=True):
        """Read frame ``idx`` with the per-particle arrays of THIS rank's partition in GPU memory.

        Restart path (BASELINE config 5): each rank reads rows ``[row0, row0 + n)`` of every
        per-particle chunk through :meth:`pgsd.fl.PGSDFile.read_chunk_device` (pread -> pinned
        slabs -> HBM -> HIP unpack); configuration, types and log stay small host values.  Chunks
        missing in the frame come from frame 0 or from the defaults, like :meth:`_read_frame`.

        Args:
            idx (int): frame index.
            part (tuple): ``(row0, n)``; default: the particles split evenly over the ranks of the
                installed communicator.
            scalar4 (bool): also assemble HOOMD-style arrays on the device:
                ``frame.particles.pos4 = (x, y, z, typeid bits)`` and ``vel4 = (vx, vy, vz, mass)``.
            defaults (bool): attributes the file holds neither in this frame nor in frame 0 are filled with their
                default rows as the host reader fills them (hoomd.py:872-881) -- views of one small device copy per
                read, a third of a 1 024-particle frame's 90 us; False leaves them ``None``.

        Returns:
            `Frame` whose ``particles.N`` is this rank's count, ``particles.N_global`` the total.  The per-particle
            arrays are torch GPU tensors where torch is importable, `pgsd.fl.DeviceBuffer` objects (the library's own
            device memory, ``__cuda_array_interface__``) otherwise -- on the GPU the file's pipeline runs on.
        """
        torch = fl._lib._torch          # None: no tensor library in this process
        if idx < 0:
            idx += len(self)
        if idx >= len(self) or idx < 0:
            raise IndexError()
        f = self.file
        snap = Frame()
        self._read_scalar_any(idx, 'configuration/step', snap.configuration, 'step')
        self._read_scalar_any(idx, 'configuration/dimensions', snap.configuration, 'dimensions')
        box_frame = idx if f.chunk_exists(idx, 'configuration/box') else (0 if f.chunk_exists(0, 'configuration/box') else None)
        if box_frame is None:
            snap.configuration.box = snap.configuration._default_value['box']
        elif box_frame == 0:
            snap.configuration.box = self._frame0_small('configuration/box').copy()
        else:
            snap.configuration.box = f.read_chunk(box_frame, 'configuration/box')

        def frame_of(chunk):
            if f.chunk_exists(idx, chunk):
                return idx
            if f.chunk_exists(0, chunk):
                return 0
            return None

        fn = frame_of('particles/N')
        n_global = 0 if fn is None else int((self._frame0_small('particles/N') if fn == 0 else f.read_chunk(fn, 'particles/N'))[0])
        ft = frame_of('particles/types')
        if ft is not None:
            tmp = self._frame0_small('particles/types') if ft == 0 else f.read_chunk(ft, 'particles/types')
            tmp = tmp.view(dtype=numpy.dtype((bytes, tmp.shape[1]))).reshape([tmp.shape[0]])
            snap.particles.types = list(a.decode('UTF-8') for a in tmp)
        else:
            snap.particles.types = snap.particles._default_value['types']

        if part is None:
            rank, size = self._comm()
            base, rem = divmod(n_global, size)
            n = base + (1 if rank < rem else 0)
            row0 = rank * base + min(rank, rem)
        else:
            row0, n = int(part[0]), int(part[1])
        snap.particles.N = n
        snap.particles.N_global = n_global
        snap.part = (row0, n)

        specs = list(_PARTICLE_SPEC.items()) + list(_PARTICLE_SPEC_EXTRA.items())
        if self._frame0_dev_part != (row0, n):
            self._frame0_dev_part, self._frame0_dev_cache = (row0, n), {}     # rows of ONE partition are kept
        cache, fresh = self._frame0_dev_cache, []
        n_frame0 = None          # frame 0's arrays stand in only while the particle count is frame 0's (hoomd.py:858-884)
        default_rows = None      # this read's own copy of the default rows (one small device-to-device copy, below)
        typed = {}
        for name, (dt, M) in specs:
            chunk = 'particles/' + name
            fr = frame_of(chunk)
            if fr == 0 and idx != 0:
                if n_frame0 is None:
                    n_frame0 = int(self._frame0_small('particles/N')[0]) if f.chunk_exists(0, 'particles/N') else n_global
                if n_frame0 != n_global:
                    fr = None
            if fr == 0 and idx != 0:
                # an array the frame does not hold because it equals frame 0's (elided by `append`): this partition's
                # rows of frame 0 are read from the file ONCE and handed out as device-to-device copies from then on
                # (a trajectory whose static arrays are elided would otherwise re-read them for every frame)
                cached = cache.get(chunk)
                if cached is not None:
                    setattr(snap.particles, name, cached.clone())
                else:
                    setattr(snap.particles, name, f.read_chunk_device(0, chunk, N=n, offset=row0, wait=False))
                    fresh.append((name, chunk))
            elif fr is not None:
                setattr(snap.particles, name, f.read_chunk_device(fr, chunk, N=n, offset=row0, wait=False))
            elif defaults and name in snap.particles._default_value:
                # like the host reader (hoomd.py:872-881) a default is ONE row broadcast over the particles: no
                # N-row allocation, no copy; `.contiguous()` / `.clone()` gives an array of its own.  The reference
                # marks its defaults read-only; torch has no such flag, so every read gets its OWN copy of the rows
                # (one clone of a 40-word template per frame, not one host->device copy per field): a write
                # through a view changes this frame's view only, never a later frame's default
                if default_rows is None:
                    default_rows = self._default_rows_template(f.pipeline_device()).clone()
                off, words, ndt = self._default_rows_layout[name]
                if torch is None:
                    # one strided view per attribute: n rows that are all the same `words` elements (stride 0)
                    view = default_rows.view(dtype=ndt, shape=(n, M) if M > 1 else (n,),
                                             strides=(0, ndt.itemsize) if M > 1 else (0,), offset_bytes=4 * off)
                else:
                    tdt = getattr(torch, ndt.name)
                    base = typed.get(tdt)
                    if base is None:
                        base = typed[tdt] = default_rows.view(tdt)  # one re-typed view of the copy per element type
                    view = base.as_strided((n, M) if M > 1 else (n,), (0, 1) if M > 1 else (0,), off)
                setattr(snap.particles, name, view)
        if scalar4 and n >= 0:
            # HOOMD's Scalar4 arrays, every row stored WHOLE by the unpack launch: the columns a missing chunk
            # would have fed come from the `fill` of the chunk that is there (type id 0 as bits, mass 1.0,
            # position / velocity 0) -- no memset of the arrays, no 12-byte stores at a 16-byte stride
            device = f.pipeline_device()
            arrays = []
            for xyz, w, w_bitcast, w_default in (('particles/position', 'particles/typeid', True, 0.0),
                                                 ('particles/velocity', 'particles/mass', False, 1.0)):
                f_xyz, f_w = frame_of(xyz), frame_of(w)
                if f_xyz is None and f_w is None:
                    # neither chunk anywhere: rows of (0, 0, 0, default w)
                    row = numpy.array([0.0, 0.0, 0.0, w_default], dtype=numpy.float32)
                    if torch is None:
                        arr = fl.DeviceBuffer((n, 4), numpy.float32, device, pattern=row)
                    else:
                        arr = torch.zeros((n, 4), dtype=torch.float32, device=torch.device('cuda', device))
                        arr[:, 3] = w_default
                elif torch is None:
                    arr = fl.DeviceBuffer((n, 4), numpy.float32, device)
                else:
                    arr = torch.empty((n, 4), dtype=torch.float32, device=torch.device('cuda', device))
                arrays.append(arr)
                if f_xyz is not None:
                    f.read_chunk_device(f_xyz, xyz, out=arr, N=n, offset=row0, columns=(0, 3), wait=False,
                                        fill=w_default if f_w is None else None)
                if f_w is not None:
                    f.read_chunk_device(f_w, w, out=arr, N=n, offset=row0, columns=(3, 4), bitcast=w_bitcast, wait=False,
                                        fill=0.0 if f_xyz is None else None)
            snap.particles.pos4, snap.particles.vel4 = arrays
        f.wait_read()
        for name, chunk in fresh:
            cache[chunk] = getattr(snap.particles, name).clone()
        for log in self._names_with_prefix('log/'):
            fr = frame_of(log)
            if fr is not None:
                snap.log[log[4:]] = f.read_chunk(fr, log)
        for state in self._names_with_prefix('state/'):
            if f.chunk_exists(idx, state):
                snap.state[state[6:]] = f.read_chunk(idx, state)
        return snap

    def _default_rows_template(self, device):
        """All default rows of the SPH schema as ONE int32 device array (every element type of the schema is four
        bytes wide), built once per trajectory and device; `_default_rows_layout[name]` = (first word, words, numpy
        dtype).  `read_frame_device` clones it per read and hands out views of the clone.  A torch tensor where torch
        is importable, the library's own device memory otherwise."""
        cached = getattr(self, '_default_rows_cache', None)
        if cached is not None and cached[0] == device:
            return cached[1]
        words, layout = [], {}
        for name, (dt, M) in _PARTICLE_SPEC.items():
            row = numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(ParticleData._default_value[name], dtype=dt), (M,)))
            layout[name] = (len(words), M, numpy.dtype(dt))
            words += row.view(numpy.int32).tolist()
        torch = fl._lib._torch
        if torch is None:
            template = fl.DeviceBuffer((len(words),), numpy.int32, device, pattern=numpy.array(words, dtype=numpy.int32))
        else:
            template = torch.tensor(words, dtype=torch.int32, device=torch.device('cuda', device))
        self._default_rows_layout = layout
        self._default_rows_cache = (device, template)
        return template

    def _frame0_small(self, chunk):
        """A small replicated chunk of frame 0 (box, counts, types ...), read once: frame 0 never changes, and a
        trajectory whose later frames fall back on it asked the file for it again for every frame read."""
        value = self._frame0_small_cache.get(chunk)
        if value is None:
            value = self.file.read_chunk(0, chunk)
            value.flags.writeable = False
            self._frame0_small_cache[chunk] = value
        return value

    def _read_scalar_any(self, idx, chunk, container, attr):
        f = self.file
        if f.chunk_exists(idx, chunk):
            setattr(container, attr, f.read_chunk(idx, chunk)[0])
        elif f.chunk_exists(0, chunk):
            setattr(container, attr, self._frame0_small(chunk)[0])
        else:
            setattr(container, attr, container._default_value[attr])

    def __getitem__(self, key):
        """``trajectory[i]`` reads one frame (negative ``i`` counts from the end), ``trajectory[a:b:c]``
        returns a lazy subset."""
        n = len(self)
        if isinstance(key, slice):
            return _FrameSubset(self, range(*key.indices(n)))
        if not isinstance(key, (int, numpy.integer)):
            raise TypeError
        idx = int(key)
        if idx < 0:
            idx += n
        if not 0 <= idx < n:
            raise IndexError()
        return self._read_frame(idx)

    def __iter__(self):
        return _FrameCursor(self, range(len(self)))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.file.close()


def open(name, mode='r'):
    """Open a hoomd schema PGSD file (hoomd.py:943-989).

    Valid modes: ``'r'``, ``'r+'``, ``'w'``, ``'x'``, ``'a'``.  Returns a `HOOMDTrajectory`.
    """
    if fl is None:
        raise RuntimeError("file layer module is not available")
    pgsdfileobj = fl.open(name=str(name), mode=mode, application='pgsd.hoomd ' + _pgsd_version,
                          schema='hoomd', schema_version=[1, 4])
    return HOOMDTrajectory(pgsdfileobj)


def read_log(name, scalar_only=False):
    """Read ``configuration/step`` and every ``log/*`` quantity into a dict of time-series arrays
    (hoomd.py:992-1075).  Quantities must keep their shape over the frames."""
    if fl is None:
        raise RuntimeError("file layer module is not available")
    with fl.open(name=str(name), mode='r', application='pgsd.hoomd ' + _pgsd_version, schema='hoomd',
                 schema_version=[1, 4]) as f:
        names = f.find_matching_chunk_names('log/')
        names.insert(0, 'configuration/step')
        if len(names) == 1:
            warnings.warn('No logged data in file: ' + str(name), RuntimeWarning)
        out = dict()
        nframes = f.nframes
        for log in names:
            exists0 = f.chunk_exists(frame=0, name=log, write_all=False)
            is_step = log == 'configuration/step'
            if exists0 or is_step:
                if is_step and not exists0:
                    tmp = numpy.array([0], dtype=numpy.uint64)
                else:
                    tmp = f.read_chunk(frame=0, name=log)
                if scalar_only and not tmp.shape[0] == 1:
                    continue
                if tmp.shape[0] == 1:
                    out[log] = numpy.full(fill_value=tmp[0], shape=(nframes,))
                else:
                    out[log] = numpy.tile(tmp, (nframes,) + tuple(1 for _ in tmp.shape))
        for idx in range(1, nframes):
            for log in out.keys():
                if not f.chunk_exists(frame=idx, name=log, write_all=False):
                    continue
                data = f.read_chunk(frame=idx, name=log)
                if len(out[log][idx].shape) == 0:
                    out[log][idx] = data[0]
                else:
                    out[log][idx] = data
    return out
