#!/bin/bash
timeout 900 python -m pytest tests/test_entry_gpu.py -q -x 2>&1 | tail -40
