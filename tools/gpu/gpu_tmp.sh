#!/bin/bash
timeout 600 python -m pytest tests/test_raymarch.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12
