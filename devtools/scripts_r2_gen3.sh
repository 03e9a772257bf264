#!/bin/bash
# leaf generation A/B, alternating to average the box's drift
for r in 1 2; do for g in 2 3; do MI355XQR_LEAF=$g ./devtools/scripts_r2_quick.sh gen$g; done; done
