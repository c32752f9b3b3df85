#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/dump_model.py /tmp/fr3_model.bin && tools/phasebench.bin /tmp/fr3_model.bin "$@"
