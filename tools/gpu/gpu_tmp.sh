#!/bin/bash
timeout 900 python -m pytest tests/test_primsdf.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
