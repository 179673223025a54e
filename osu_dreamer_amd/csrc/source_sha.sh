#!/bin/bash
# 16 hex digits identifying the kernel sources a library was built from (and the extra hipcc flags): every *.hip / *.h here plus the
# public header.  Baked into the library as od_build_source_sha(); profiles and parity records quote it, bench.py drops any record whose
# hash differs from the library it runs.
cd "$(dirname "$0")"
{ for f in $(ls *.hip *.h | LC_ALL=C sort); do echo "== $f"; cat "$f"; done; echo "== include"; cat ../../include/osu_dreamer_hip.h; echo "== flags ${OD_HIPCC_FLAGS}"; } | sha256sum | cut -c1-16
