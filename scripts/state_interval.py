#!/usr/bin/env python3
"""JSON counts -> state estimate + HS confidence radii (same interface as the reference's script)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantpy_amd.cli import state_interval  # noqa: E402

if __name__ == "__main__":
    state_interval()
