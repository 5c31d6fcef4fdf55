#!/bin/bash
# Stage (or remove) a scratch copy of the reference checkout INSIDE the repo so that ONE gpurun session can execute the
# reference's unmodified eval.py on the MI355X (the GPU box has no /root/reference; gpurun ships the repo snapshot).
#   bash tools/stage_reference.sh          copy /root/reference (minus web/, .git, caches) -> _scratch_reference/
#   bash tools/stage_reference.sh clean    delete it again  (ALWAYS after the session: nothing of the reference stays in the tree)
# _scratch_reference/ is git-ignored and never committed.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=$ROOT/_scratch_reference
if [ "$1" = "clean" ]; then rm -rf "$D"; echo "removed $D"; exit 0; fi
rm -rf "$D"; mkdir -p "$D"
(cd /root/reference && tar --exclude=web --exclude=.git --exclude=__pycache__ --exclude='*.pth' -cf - .) | (cd "$D" && tar xf -)
du -sh "$D"
