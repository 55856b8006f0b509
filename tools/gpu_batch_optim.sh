#!/bin/bash
timeout 900 python -m pytest tests/test_sampler_features_gpu.py tests/test_conditional_gpu.py -x -q -m gpu 2>&1 | tail -2
