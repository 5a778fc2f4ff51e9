#!/bin/bash
# The compile-time variants of round 6 that wait for a timing on the GPU (profiles/r06_notes.md), rebuilt from the current source:
# exp_libs/libplmpm_NAME.so for tools/replay_ab.py / tools/ab.py.  Parity of the source variants themselves (not of these gfx950
# builds) is checked on the CPU interpreter: PLMPM_EMUL_VARIANT=bufio|pk|pkbuf, tests/test_emul_tier.py.
set -e
cd "$(dirname "$0")"
./build_variant.sh bufio    "-DPLB_BUFIO=1"
./build_variant.sh g3       "-DPLB_P2G_GRAD_WAVES=3"
./build_variant.sh g3buf    "-DPLB_P2G_GRAD_WAVES=3 -DPLB_BUFIO=1"
./build_variant.sh g3bufpk2 "-DPLB_P2G_GRAD_WAVES=3 -DPLB_BUFIO=1 -DPLB_PK_GATHER=2"
./build_variant.sh pk1      "-DPLB_PK_GATHER=1"
./build_variant.sh pk1buf   "-DPLB_PK_GATHER=1 -DPLB_BUFIO=1"
./build_variant.sh pk3      "-DPLB_PK_GATHER=3"
./build_variant.sh pk3buf   "-DPLB_PK_GATHER=3 -DPLB_BUFIO=1"
./build_variant.sh d2       "-DPLB_P2G_WAVES_F64=2"
