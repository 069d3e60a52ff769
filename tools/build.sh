#!/bin/bash
# make the library and FAIL LOUDLY (a failed audit deletes the object and leaves the previous .so in place: never measure that by accident)
set -e
cd "$(dirname "$0")/.."
make -C retrieval-augmented-diffusion-models_amd/csrc -j8 > /tmp/rdm_build.log 2>&1 || { grep -E "error|Error|check_" /tmp/rdm_build.log | head -20; echo "BUILD FAILED"; exit 1; }
echo "build ok: $(ls -la retrieval-augmented-diffusion-models_amd/librdm_hip.so | awk '{print $6, $7, $8}')"
