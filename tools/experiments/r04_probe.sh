#!/bin/bash
# probe.sh NT "KEEP PHASES" [extra flags]
NT=$1; keep="$2"; shift; shift
ALLK="CHUNKS IMU ASM CHOL BACKSUB COST SETUP"; fl="-DUVS_X_NO_REDAMP"
for k in $ALLK; do case " $keep " in *" $k "*) ;; *) fl="$fl -DUVS_XK_$k";; esac; done
r=$($(dirname $0)/r04_phase_probe.sh $NT $fl "$@"); echo "NT=$NT keep [$keep] $*: $r"
