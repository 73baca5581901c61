"""Bit-equality of two builds of libhippt.so: python tools/ab_equal.py A.so B.so  (each rendered in its own process)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, hashlib; sys.path.insert(0, %r)
from pathtracer_amd import Scene, hip_renderer
for name, w, h, spp in (("scene.example.json", 333, 201, 40), ("glass_fog.json", 256, 144, 24), ("scene.example.json", 1920, 1080, 8)):
    s = Scene.load(%r + "/scenes/" + name, w, h, spp)
    print(name, w, h, spp, hashlib.sha1(hip_renderer.render_accum(s, seed=11).tobytes()).hexdigest())
""" % (ROOT, ROOT)
outs = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, HIPPT_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stderr); sys.exit(1)
    outs.append(r.stdout)
print(outs[0]); print("EQUAL" if outs[0] == outs[1] else "DIFFERENT\n" + outs[1])
