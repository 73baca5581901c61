"""Authors the benchmark scenes that are not in the reference.

glass_fog.json (BASELINE config C3): scene.example.json's geometry with the
glass material changed to ior=1.5, volume_density=0.5, volume_anisotropy=0.3,
rgb=(0.9, 0.9, 0.95).
Run: python scenes/make_scenes.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    with open(os.path.join(HERE, "scene.example.json")) as f:
        sc = json.load(f)
    g = sc["materials"]["Glass.006"]
    g["rgb"] = [0.9, 0.9, 0.95]
    g["ior"] = 1.5
    g["volume_density"] = 0.5
    g["volume_anisotropy"] = 0.3
    sc["render"] = {"width": 1920, "height": 1080, "samples": 4096}
    with open(os.path.join(HERE, "glass_fog.json"), "w") as f:
        json.dump(sc, f, indent=4)
        f.write("\n")


if __name__ == "__main__":
    main()
