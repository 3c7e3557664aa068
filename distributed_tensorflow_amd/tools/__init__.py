"""Command-line tools (``python -m distributed_tensorflow_amd.tools.<name>``)."""
