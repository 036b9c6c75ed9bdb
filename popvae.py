#!/usr/bin/env python3
"""Drop-in for the reference's ``popvae.py`` command line (same flags, same output files), running the
MI355X engine.  See popvae_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from popvae_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
