#!/bin/bash
# final refresh of round 2 after the LDS swizzle + measured tile selection: full suite, smoke, bench, kernel stats, PMC passes, A/B vs the padded build
SKIP_CFG3=1 SKIP_TRAIN=1 bash tools/refresh_profiles.sh r02f
