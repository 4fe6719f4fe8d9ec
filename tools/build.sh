#!/bin/bash
# quiet in-tree build of libafhip.so (prints only errors and the last line)
cd "$(dirname "$0")/.." && python3 -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','spatiotemporal-deepfake-detection-for-live-video-calls_amd/build.py');m=importlib.util.module_from_spec(spec);spec.loader.exec_module(m);print(m.build(verbose=False))"
