#!/bin/bash
# second half of tools/r04_profiles.sh: steps 4-6
{ sed -n '/^set -o pipefail/,/^step() /p' tools/r04_profiles.sh; sed -n '/^step "4 PMC/,$p' tools/r04_profiles.sh; } > /tmp/r04_b.sh; bash /tmp/r04_b.sh
