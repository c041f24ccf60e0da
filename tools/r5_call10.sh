#!/bin/bash
set -o pipefail
for a in "65 8" "65 2" "129 8" "129 2" "25 8" "25 2"; do timeout -k 5 60 ./tools/walklab/walklab $a 5 || exit 1; done
