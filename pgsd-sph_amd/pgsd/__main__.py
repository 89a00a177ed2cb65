"""Command line of the package: ``python -m pgsd <command>``.

``read``   the reference's one command (``__main__.py:52-87``): an interactive Python prompt with
           the file open as ``handle`` and, for the hoomd schema, the trajectory as ``traj``.
``info``   header, frame count and the chunks of one frame, printed and done (no prompt).
``vtu``    every frame as a VTK ``.vtu`` file plus a ``.pvd`` collection (``pgsd.vtu``).
"""
import argparse
import code
import sys

from .version import __version__

_MODES = ['rb', 'rb+', 'wb', 'wb+', 'xb', 'xb+', 'ab', 'w', 'r', 'r+', 'x', 'a']

_BANNER = """Python {python}
pgsd {version}

File: {name}
{extras}
Variables: "handle" is the open pgsd.fl.PGSDFile; with the hoomd schema "traj" is the
pgsd.hoomd.HOOMDTrajectory on top of it. The modules pgsd, pgsd.fl and pgsd.hoomd are imported.
help(handle) and help(traj) describe them."""


def _cmd_read(args):
    import pgsd
    from . import fl, hoomd
    ns = {'pgsd': pgsd, 'pgsd.fl': fl, 'pgsd.hoomd': hoomd}
    extras = []
    if args.schema == 'hoomd':
        traj = hoomd.open(args.file, mode=args.mode)
        ns['traj'] = traj
        ns['handle'] = traj.file
        extras.append("Number of frames: %d" % len(traj))
    else:
        if args.mode not in ('rb', 'rb+', 'ab', 'a', 'r', 'r+'):
            raise ValueError("Unsupported schema for creating a file.")
        ns['handle'] = fl.open(args.file, args.mode)
    code.interact(local=ns, banner=_BANNER.format(python=sys.version, version=__version__, name=args.file,
                                                  extras="\n".join(extras) + "\n"))


def _cmd_info(args):
    from . import fl
    with fl.open(args.file, 'r') as f:
        print("file:            %s" % args.file)
        print("application:     %s" % f.application)
        print("schema:          %s %d.%d" % ((f.schema,) + tuple(f.schema_version)))
        print("pgsd version:    %d.%d" % tuple(f.pgsd_version))
        print("frames:          %d" % f.nframes)
        print("chunk names:     %d" % f.nnames)
        if f.nframes == 0:
            return
        frame = args.frame if args.frame >= 0 else f.nframes + args.frame
        if not 0 <= frame < f.nframes:
            raise ValueError("frame %d is not in the file" % args.frame)
        print("chunks of frame %d:" % frame)
        for name in f.find_matching_chunk_names(''):
            if f.chunk_exists(frame, name):
                data = f.read_chunk(frame, name)
                print("  %-28s %-8s %s" % (name, data.dtype, 'x'.join(str(n) for n in data.shape)))


def _cmd_vtu(args):
    from . import vtu
    for name in vtu.pgsd2vtu(args.file, args.output):
        print(name)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    parser = argparse.ArgumentParser(prog="pgsd", description="Readers and writers of the PGSD (GSD v2) file format.")
    parser.add_argument('--version', action='store_true', help="Display the version number and exit.")
    parser.add_argument('--debug', action='store_true', help="Show traceback on error for debugging.")
    sub = parser.add_subparsers()
    p = sub.add_parser('read', help="interactive prompt with the file open")
    p.add_argument('file', type=str, nargs='?', help="PGSD file to read.")
    p.add_argument('-s', '--schema', type=str, default='hoomd', choices=['hoomd', 'none'], help="The file schema.")
    p.add_argument('-m', '--mode', type=str, default='r', choices=_MODES, help="The file mode.")
    p.set_defaults(func=_cmd_read)
    p = sub.add_parser('info', help="print header and chunk list")
    p.add_argument('file', type=str)
    p.add_argument('-f', '--frame', type=int, default=-1, help="frame whose chunks are listed (default: last)")
    p.set_defaults(func=_cmd_info)
    p = sub.add_parser('vtu', help="convert the frames to VTK .vtu files")
    p.add_argument('file', type=str)
    p.add_argument('-o', '--output', type=str, default=None, help="output directory (default: next to the file)")
    p.set_defaults(func=_cmd_vtu)

    if '--version' in argv:  # works without a subcommand, like the reference (__main__.py:139-145)
        print('pgsd', __version__)
        return 0
    args = parser.parse_args(argv)
    if not hasattr(args, 'func'):
        parser.print_usage()
        return 2
    try:
        args.func(args)
    except KeyboardInterrupt:
        print("\nInterrupted.", file=sys.stderr)
        if args.debug:
            raise
        return 1
    except Exception as error:  # noqa: BLE001 - the command line reports, --debug re-raises
        print('Error: {}'.format(error), file=sys.stderr)
        if args.debug:
            raise
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
