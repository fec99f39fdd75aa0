#!/bin/bash
# Resource usage of every kernel of both translation units: name, VGPRs, SGPRs, scratch bytes/lane, occupancy, LDS.
# usage: tools/resusage.sh [extra make args]
cd "$(dirname "$0")/../renderer_amd/csrc" && make asm "$@" 2>&1 | awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
/TotalSGPRs:/ {sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg)}
/ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
/Occupancy/ {oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc)}
/LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-110s vgpr %3s sgpr %3s scratch %4s occ %s lds %s\n", name, v, sg, sc, oc, l}
/error:/ {print}' | sort -u
