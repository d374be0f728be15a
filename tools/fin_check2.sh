#!/bin/bash
# split (default) and in-order k_fin side by side
for e in 0 1; do
  if [ $e = 1 ]; then export GRAAL_NO_EARLY_FIN=1; echo "#### GRAAL_NO_EARLY_FIN=1"; else unset GRAAL_NO_EARLY_FIN; echo "#### default"; fi
  tools/fin_check.sh || exit 1
done
