#!/usr/bin/env python3
"""Attribute names of the reference's module classes, extracted by AST from the files where they lie
under /root/reference (nothing is imported: mmcv / mmdet are absent).  For every class of interest the
set of names assigned as `self.<name> = ...` or registered with `self.add_module("<name>", ...)` in any
method, plus names added to a child (`self.can_bus_mlp.add_module("norm", ...)`).  The checkpoint key
map (bevformer_tensorrt_amd/checkpoint.py) is tested against these names
(tests/test_checkpoint_cpu.py).  Writes tests/golden/reference_module_names.json."""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = {
    "det2trt/models/dense_heads/bevformer_head.py": None,
    "det2trt/models/modules/transformer.py": None,
    "det2trt/models/modules/spatial_cross_attention.py": None,
    "det2trt/models/modules/temporal_self_attention.py": None,
    "det2trt/models/modules/decoder.py": None,
    "det2trt/models/modules/encoder.py": None,
    "det2trt/models/backbones/resnet.py": None,
    "det2trt/models/modules/cnn/dcn.py": None,
    "third_party/bev_mmdet3d/models/necks/fpn.py": None,
    "det2trt/models/detector/bevformer.py": None,
}


def names_of(cls):
    out = set()
    for node in ast.walk(cls):
        if isinstance(node, (ast.Assign, ast.AnnAssign)):
            targets = node.targets if isinstance(node, ast.Assign) else [node.target]
            for t in targets:
                for tt in (t.elts if isinstance(t, ast.Tuple) else [t]):
                    if isinstance(tt, ast.Attribute) and isinstance(tt.value, ast.Name) and tt.value.id == "self":
                        out.add(tt.attr)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_module":
            if node.args and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str):
                owner = node.func.value
                prefix = ""
                if isinstance(owner, ast.Attribute) and isinstance(owner.value, ast.Name) and owner.value.id == "self":
                    prefix = owner.attr + "."
                out.add(prefix + node.args[0].value)
            elif node.args and isinstance(node.args[0], ast.JoinedStr):
                # f"layer{i + 1}" -> "layer{}"
                out.add("".join(v.value if isinstance(v, ast.Constant) else "{}" for v in node.args[0].values))
    return sorted(out)


def main():
    res = {}
    for rel in FILES:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef):
                res[f"{rel}::{node.name}"] = {"bases": [ast.unparse(b) for b in node.bases], "names": names_of(node)}
    with open(os.path.join(HERE, "reference_module_names.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(len(res), "classes")


if __name__ == "__main__":
    main()
