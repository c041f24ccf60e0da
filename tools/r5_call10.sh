#!/bin/bash
set -o pipefail
for a in "65 8" "129 8" "25 8"; do timeout -k 5 60 ./tools/walklab/walklab $a 5 || exit 1; done
