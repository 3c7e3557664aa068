"""``python -m distributed_tensorflow_amd <command> ...`` -- one front door for the pieces that have a command line.

    serve [--all-in-one] [--gpus 0,1,..]     the REST service (``service/disTf.py``; the reference's ``disTf.py``)
    cleaner | reaper                         the watcher and the ps reaper as separate daemons (``tf_cleaner.py`` / ``remove_ps.py``)
    job submit|status|wait|cancel|logs ...   client of ``/v1/train`` (``tools/job.py``)
    inspect <export> ...                     what a job left in its export directory (``tools/inspect_checkpoint.py``)
    evaluate --model M --export E --data D   score a checkpoint on a dataset directory (``tools/evaluate.py``)
    build [--debug-bounds]                   compile the HIP kernel library and the native runtime for gfx950
"""
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help", "help"):
        print(__doc__.strip())
        return 0
    cmd, rest = argv[0], argv[1:]
    if cmd == "serve":
        from .service import disTf

        return disTf.main(rest) or 0
    if cmd == "cleaner":
        from .service import tf_cleaner

        sys.argv = ["tf_cleaner"] + rest
        return tf_cleaner.main() or 0
    if cmd == "reaper":
        from .service import remove_ps

        sys.argv = ["remove_ps"] + rest
        return remove_ps.main() or 0
    if cmd == "job":
        from .tools import job

        return job.main(rest)
    if cmd == "inspect":
        from .tools import inspect_checkpoint

        return inspect_checkpoint.main(rest)
    if cmd == "evaluate":
        from .tools import evaluate

        return evaluate.main(rest)
    if cmd == "build":
        import runpy

        sys.argv = ["build"] + rest
        runpy.run_module("distributed_tensorflow_amd.ops.build", run_name="__main__")
        if "--debug-bounds" not in rest:
            sys.argv = ["build"]
            runpy.run_module("distributed_tensorflow_amd.native.build", run_name="__main__")
        print("build: up to date")
        return 0
    print("unknown command %r\n\n%s" % (cmd, __doc__.strip()), file=sys.stderr)
    return 2


if __name__ == "__main__":
    sys.exit(main())
