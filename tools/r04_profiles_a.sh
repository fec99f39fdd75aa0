#!/bin/bash
# first half of tools/r04_profiles.sh (a gpurun call is limited to 20 minutes): steps 1-3
sed -n '/^set -o pipefail/,/^step "4 PMC/p' tools/r04_profiles.sh | sed '$d' > /tmp/r04_a.sh; bash /tmp/r04_a.sh
