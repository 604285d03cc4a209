#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
