"""Read-only, pure-Python reader of PGSD / GSD files.

Counterpart of the reference's ``pgsd.pypgsd`` (/root/reference/pgsd/pgsd/pypgsd.py:70-433):
same class name, methods and properties, interchangeable with :class:`pgsd.fl.PGSDFile` for
reading (``HOOMDTrajectory(PGSDFile(open(path, 'rb')))``).  It needs neither the shared
library nor a GPU, which makes it the independent check that files written by the device
path are well-formed.  The header and the whole index block are decoded with numpy
structured dtypes in one shot instead of one ``struct.unpack`` per entry, and version-2
files are searched by bisection on ``(frame, id)``.
"""
import logging
import sys

import numpy

version = "3.2.0"

logger = logging.getLogger('pgsd.pypgsd')

# on-disk layouts, pgsd.h:143-204 (pypgsd.py:43-54 spells them 'QQQQQII64s64s80s' / 'QQqIHBB')
_header_dtype = numpy.dtype([('magic', '<u8'), ('index_location', '<u8'), ('index_allocated_entries', '<u8'),
                             ('namelist_location', '<u8'), ('namelist_allocated_entries', '<u8'),
                             ('schema_version', '<u4'), ('pgsd_version', '<u4'),
                             ('application', 'S64'), ('schema', 'S64'), ('reserved', 'S80')])
_index_dtype = numpy.dtype([('frame', '<u8'), ('N', '<u8'), ('location', '<i8'), ('M', '<u4'), ('id', '<u2'),
                            ('type', 'u1'), ('flags', 'u1')])
assert _header_dtype.itemsize == 256 and _index_dtype.itemsize == 32

pgsd_type_mapping = {
    1: numpy.dtype('uint8'), 2: numpy.dtype('uint16'), 3: numpy.dtype('uint32'), 4: numpy.dtype('uint64'),
    5: numpy.dtype('int8'), 6: numpy.dtype('int16'), 7: numpy.dtype('int32'), 8: numpy.dtype('int64'),
    9: numpy.dtype('float32'), 10: numpy.dtype('float64'),
}

_MAGIC = 0x65DF65DF65DF65DF


class PGSDFile(object):
    """PGSD file access interface over any binary file-like object (read-only).

    Args:
        file: file-like object opened in binary mode.
    """

    def __init__(self, file):
        self.__file = file
        self.__is_open = False
        logger.info('opening file: ' + str(file))

        self.__file.seek(0)
        try:
            raw = self.__file.read(_header_dtype.itemsize)
        except UnicodeDecodeError:
            print("\nDid you open the file in binary mode (rb)?\n", file=sys.stderr)
            raise
        if len(raw) != _header_dtype.itemsize:
            raise IOError
        self.__header = numpy.frombuffer(raw, dtype=_header_dtype)[0]

        # validate the header (pypgsd.py:123-132)
        if int(self.__header['magic']) != _MAGIC:
            raise RuntimeError("Not a PGSD file: " + str(self.__file))
        v = int(self.__header['pgsd_version'])
        if (v < (1 << 16) and v != (0 << 16 | 3)) or v >= (3 << 16):
            raise RuntimeError("Unsupported PGSD file version: " + str(self.__file))
        self.__v1 = v < (2 << 16)

        # names: first-seen order defines the ids
        self.__file.seek(int(self.__header['namelist_location']), 0)
        names_raw = self.__file.read(int(self.__header['namelist_allocated_entries']) * 64)
        self.__namelist = {}
        if self.__v1:
            chunks = [names_raw[i:i + 64].split(b'\x00', 1)[0] for i in range(0, len(names_raw), 64)]
        else:
            chunks = names_raw.split(b'\x00')
        for chunk in chunks:
            if len(chunk) == 0:
                if self.__v1:
                    break
                continue
            name = chunk.decode('utf-8')
            self.__namelist.setdefault(name, len(self.__namelist))

        # index: one read, used prefix = entries before the first location == 0
        n_alloc = int(self.__header['index_allocated_entries'])
        self.__file.seek(int(self.__header['index_location']), 0)
        raw = self.__file.read(n_alloc * _index_dtype.itemsize)
        if len(raw) != n_alloc * _index_dtype.itemsize:
            raise IOError
        index = numpy.frombuffer(raw, dtype=_index_dtype)
        empty = numpy.nonzero(index['location'] == 0)[0]
        used = int(empty[0]) if empty.size else n_alloc
        index = index[:used]
        if used:
            ok = numpy.isin(index['type'], list(pgsd_type_mapping)) & (index['M'] != 0) \
                & (index['frame'] < n_alloc) & (index['id'] < len(self.__namelist)) & (index['flags'] == 0)
            if not ok.all() or (numpy.diff(index['frame'].astype(numpy.int64)) < 0).any():
                raise RuntimeError("Corrupt PGSD file: " + str(self.__file))
        self.__index = index
        # sort key for bisection in v2 files (the writer keeps the index sorted by (frame, id))
        self.__keys = (index['frame'].astype(numpy.uint64) << numpy.uint64(16)) | index['id'].astype(numpy.uint64)
        self.__is_open = True

    def close(self):
        """Close the file; may be called more than once."""
        if self.__is_open:
            logger.info('closing file: ' + str(self.__file))
            self.__index = None
            self.__namelist = None
            self.__is_open = False
            self.__file.close()

    def end_frame(self):
        """Not implemented (read-only)."""
        raise NotImplementedError

    def write_chunk(self, name, data):
        """Not implemented (read-only)."""
        raise NotImplementedError

    def _find_chunk(self, frame, name):
        match_id = self.__namelist.get(name)
        if match_id is None or len(self.__index) == 0:
            return None
        if not self.__v1 and int(frame) < (1 << 47):
            key = (numpy.uint64(frame) << numpy.uint64(16)) | numpy.uint64(match_id)
            pos = int(numpy.searchsorted(self.__keys, key))
            if pos < len(self.__keys) and self.__keys[pos] == key:
                return self.__index[pos]
            # an index that is only frame-ordered falls through to the scan below
        frames = self.__index['frame']
        lo = int(numpy.searchsorted(frames, frame, side='left'))
        hi = int(numpy.searchsorted(frames, frame, side='right'))
        # the reference scans from the last entry of the frame backwards (pypgsd.py:247-253)
        for i in range(hi - 1, lo - 1, -1):
            if int(self.__index[i]['id']) == match_id:
                return self.__index[i]
        return None

    def chunk_exists(self, frame, name, write_all=False):
        """True if the chunk exists in the file at the given frame."""
        if not self.__is_open:
            raise ValueError("File is not open")
        return self._find_chunk(frame, name) is not None

    def read_chunk(self, frame, name, offset=0, r_all=False):
        """Read a data chunk and return it as a numpy array ((N,) for Nx1, else (N, M)).

        ``offset`` and ``r_all`` are accepted for interface parity and ignored: the whole
        chunk is returned (pypgsd.py:284-347)."""
        if not self.__is_open:
            raise ValueError("File is not open")
        chunk = self._find_chunk(frame, name)
        if chunk is None:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + str(self.__file))
        logger.debug('read chunk: ' + str(self.__file) + ' - ' + str(frame) + ' - ' + name)
        dt = pgsd_type_mapping[int(chunk['type'])]
        N, M = int(chunk['N']), int(chunk['M'])
        size = N * M * dt.itemsize
        if int(chunk['location']) == 0:
            raise RuntimeError("Corrupt chunk: " + str(frame) + " / " + name + " in file" + str(self.__file))
        if size == 0:
            return numpy.array([], dtype=dt)
        self.__file.seek(int(chunk['location']), 0)
        raw = self.__file.read(size)
        if len(raw) != size:
            raise IOError
        data = numpy.frombuffer(raw, dtype=dt)
        return data if M == 1 else data.reshape([N, M])

    def find_matching_chunk_names(self, match, write_all=False):
        """Chunk names in the file that start with ``match``."""
        return [key for key in self.__namelist.keys() if key.startswith(match)]

    def __getstate__(self):
        return dict(name=self.name)

    def __setstate__(self, state):
        self.__init__(open(state['name'], 'rb'))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    @property
    def name(self):
        """(str): file.name."""
        return self.__file.name

    @property
    def file(self):
        """File-like object opened."""
        return self.__file

    @property
    def mode(self):
        """str: Mode of the open file."""
        return 'r'

    @property
    def pgsd_version(self):
        """tuple[int, int]: PGSD file layer version (major, minor)."""
        v = int(self.__header['pgsd_version'])
        return (v >> 16, v & 0xffff)

    @property
    def schema_version(self):
        """tuple[int, int]: schema version (major, minor)."""
        v = int(self.__header['schema_version'])
        return (v >> 16, v & 0xffff)

    @property
    def schema(self):
        """str: name of the data schema."""
        return bytes(self.__header['schema']).rstrip(b'\x00').decode('utf-8')

    @property
    def application(self):
        """str: name of the generating application."""
        return bytes(self.__header['application']).rstrip(b'\x00').decode('utf-8')

    @property
    def nframes(self):
        """int: number of frames in the file."""
        if not self.__is_open:
            raise ValueError("File is not open")
        if len(self.__index) == 0:
            return 0
        return int(self.__index[-1]['frame']) + 1
