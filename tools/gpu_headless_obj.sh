#!/bin/bash
# GPU check of the C++ facade's OBJ-in / PPM-out path with the headless driver
set -e
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, rt_amd as ra
scene, _ = ra.Scene.builtin(1, param=5)
with open('/tmp/mesh.obj', 'w') as f:
    for i, t in enumerate(scene.triangles):
        for v in (t['v0'], t['v1'], t['v2']):
            f.write('v %r %r %r\n' % tuple(float(c) for c in v))
        f.write('f %d %d %d\n' % (3 * i + 1, 3 * i + 2, 3 * i + 3))
print('wrote /tmp/mesh.obj', len(scene.triangles), 'triangles')
PY
timeout -k 10 120 rt_amd/host/rt_headless /tmp/mesh.obj 640 360 8 8 0 gpurun_out/headless_obj.ppm
python3 - <<'PY'
import numpy as np
raw = open('gpurun_out/headless_obj.ppm', 'rb').read()
assert raw.startswith(b'P6\n640 360\n255\n')
px = np.frombuffer(raw[len(b'P6\n640 360\n255\n'):], np.uint8).reshape(360, 640, 3)
print('ppm ok: mean rgb', px.reshape(-1, 3).mean(0).round(1), 'distinct colours', len(np.unique(px.reshape(-1, 3), axis=0)))
PY
