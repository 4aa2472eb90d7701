"""INTEGRATION.md shows the Rust declarations a maintainer of the reference would add (no Rust toolchain
exists here to compile them). This keeps them honest: every #[repr(C)] struct in the document must list
the same fields, with the same types, in the same order as the C struct in include/portrayer_hip.h,
and every extern "C" function it declares must exist in the header with the same number of arguments."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "int": "c_int", "float": "f32",
             "pt_rect": "PtRect"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_structs(header):
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(\w+);", strip_comments(header), flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const )?(\w+) (.*)", decl)
            const, base, rest = m.group(1), m.group(2), m.group(3)
            for item in rest.split(","):
                item = item.strip()
                ptr = item.startswith("*")
                item = item.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]", item)
                if arr:
                    fields.append((arr.group(1), f"[{C_SCALARS[base]}; {arr.group(2)}]"))
                elif ptr:
                    fields.append((item, ("*const " if const else "*mut ") + C_SCALARS[base]))
                else:
                    fields.append((item, C_SCALARS[base]))
        out[name] = fields
    return out


def rust_structs(doc):
    out = {}
    for name, body in re.findall(r"pub struct (\w+)\s*\{(.*?)\n?\}", doc, flags=re.S):
        body = re.sub(r"//[^\n]*", "", body)
        out[name] = [(f, " ".join(t.split())) for f, t in re.findall(r"pub (\w+):\s*([^,]+?)\s*(?:,|$)", body, flags=re.S)]
    return out


def test_rust_structs_match_the_c_header():
    header = open(os.path.join(ROOT, "include", "portrayer_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    c, r = c_structs(header), rust_structs(doc)
    pairs = {"pt_scene": "PtScene", "pt_kdtree": "PtKdTree", "pt_camera": "PtCamera", "pt_rect": "PtRect",
             "pt_render_params": "PtRenderParams", "pt_stats": "PtStats"}
    for cname, rname in pairs.items():
        assert cname in c and rname in r, (cname, rname)
        assert r[rname] == c[cname], f"{rname} differs from {cname}:\n rust {r[rname]}\n c    {c[cname]}"


def test_rust_extern_functions_exist_in_the_header():
    header = strip_comments(open(os.path.join(ROOT, "include", "portrayer_hip.h")).read())
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', doc, flags=re.S).group(1)
    fns = re.findall(r"pub fn (\w+)\((.*?)\)", block, flags=re.S)
    assert len(fns) >= 7
    for name, args in fns:
        m = re.search(r"\b" + name + r"\((.*?)\);", header, flags=re.S)
        assert m, f"{name} is not declared in portrayer_hip.h"
        n_c = 0 if m.group(1).strip() == "void" else m.group(1).count(",") + 1
        assert args.count(":") == n_c, f"{name}: {args.count(':')} arguments in INTEGRATION.md, {n_c} in the header"
