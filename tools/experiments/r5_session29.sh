#!/bin/bash
# round 5, session 29: context reuse for shorter clips -- the state tests and the rest of the suite
cd $(pwd)
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
