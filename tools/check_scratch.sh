#!/bin/bash
# Scratch (private-memory) traffic inside the MFMA loops of a built library: llvm-objdump of the gfx950 code objects, per kernel the scratch_* instructions
# inside its innermost MFMA loops (must be 0), between its first and last MFMA, and in total.  See tools/check_scratch.py for the rules and the allow-list.
#   tools/check_scratch.sh [library.so] [--all] [--json]        (default: motionrag_amd/libmrag_hip.so; exit code 1 = scratch inside a hot MFMA loop)
exec python3 "$(dirname "$0")/check_scratch.py" "$@"
