#!/usr/bin/env python3
"""CLI of trajectory_optimization_matrix_lie_groups_amd/_dpp_lint.py: python tools/dpp_hazard_lint.py [lib.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd._dpp_lint import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
