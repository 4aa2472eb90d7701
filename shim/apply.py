#!/usr/bin/env python3
"""Makes a copy of a portrayer checkout whose `ImageSliceMut::render` runs on an MI355X (cargo feature `hip`).

    python3 shim/apply.py /path/to/portrayer /path/to/portrayer-hip
    cd /path/to/portrayer-hip
    PORTRAYER_HIP_DIR=/path/to/this/repo/portrayer_amd cargo run --release --features hip --example big-scene

What it does to the copy - nothing is replaced, the crate's own code stays as it is:
  * adds build.rs, src/hip_ffi.rs, src/hip_pack.rs (this directory);
  * appends read-only `hip_*` accessors to six modules (shim/overlay/*.rs.append): hip_pack lives in another module and
    cannot see their private fields;
  * appends `render_hip` to src/render.rs and inserts ONE line at the top of the body of `ImageSliceMut::render` that
    delegates to it when the feature is on;
  * declares the modules in src/lib.rs and the feature + build script in Cargo.toml.
Without `--features hip` the crate builds and behaves exactly as before.

This environment has no Rust toolchain: the overlay has NOT been compiled here. tests/shim_replay.c replays the call
sequence and the array layouts of hip_pack.rs / render_hip through libportrayer_hip.so on every GPU test run."""
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
APPENDS = {"src/primitive/mesh.rs": "primitive_mesh.rs.append", "src/kdtree/kdmesh.rs": "kdtree_kdmesh.rs.append",
           "src/kdtree/node.rs": "kdtree_node.rs.append", "src/bounding_box.rs": "bounding_box.rs.append",
           "src/camera.rs": "camera.rs.append", "src/texture.rs": "texture.rs.append", "src/render.rs": "render.rs.append"}


def main(src, dst):
    if os.path.exists(dst):
        sys.exit(f"{dst} exists")
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("target", ".git"))
    for f in ("build.rs",):
        shutil.copy(os.path.join(HERE, f), os.path.join(dst, f))
    for f in ("hip_ffi.rs", "hip_pack.rs"):
        shutil.copy(os.path.join(HERE, "src", f), os.path.join(dst, "src", f))
    for target, frag in APPENDS.items():
        with open(os.path.join(dst, target), "a") as out, open(os.path.join(HERE, "overlay", frag)) as add:
            out.write(add.read())
    # the delegation: first line of the body of `pub fn render<..>(&mut self, scene: &HierScene, ..)` of ImageSliceMut
    p = os.path.join(dst, "src", "render.rs")
    text = open(p).read()
    m = re.search(r"impl<'a> ImageSliceMut<'a> \{.*?pub fn render<[^{]*?background: T,\s*\) \{\n", text, re.S)
    if not m:
        sys.exit("src/render.rs: ImageSliceMut::render not found where expected (render.rs:93-98)")
    text = text[:m.end()] + "        #[cfg(feature = \"hip\")]\n        { return self.render_hip::<R, T>(scene, camera, background); }\n" + text[m.end():]
    open(p, "w").write(text)
    # modules
    p = os.path.join(dst, "src", "lib.rs")
    with open(p, "a") as out:
        out.write("\n#[cfg(feature = \"hip\")]\nmod hip_ffi;\n#[cfg(feature = \"hip\")]\nmod hip_pack;\n")
    # Cargo.toml: the feature and the build script
    p = os.path.join(dst, "Cargo.toml")
    text = open(p).read()
    if "[features]" in text:
        text = text.replace("[features]", "[features]\n# ImageSliceMut::render on an MI355X through libportrayer_hip.so (PORTRAYER_HIP_DIR)\nhip = []", 1)
    else:
        text += "\n[features]\n# ImageSliceMut::render on an MI355X through libportrayer_hip.so (PORTRAYER_HIP_DIR)\nhip = []\n"
    text = re.sub(r"(\[package\]\n)", r'\1build = "build.rs"\n', text, count=1)
    open(p, "w").write(text)
    print(f"wrote {dst}: build with  PORTRAYER_HIP_DIR=<dir of libportrayer_hip.so> cargo build --release --features hip")


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    main(sys.argv[1], sys.argv[2])
