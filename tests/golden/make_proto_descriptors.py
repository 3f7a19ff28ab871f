#!/usr/bin/env python3
"""Builds tests/golden/comet_protos.desc — a serialized google.protobuf FileDescriptorSet of the reference's wire schema
(native/proto/src/proto/{types,literal,expr,partitioning,operator,config,metric}.proto) — WITHOUT protoc: a small proto3 parser
(messages, enums, oneofs, nested types, map<,>, optional, reserved, imports) turns the .proto text into descriptors.

The descriptor set is the independent referee between the two hand-written codecs of this repo (serde.py encodes, csrc/proto.cpp
decodes): tests/test_proto_wire_cpu.py parses serde's bytes with google.protobuf built from these descriptors and fails on any
field this schema does not know, and feeds protobuf's own re-serialization to proto.cpp.

Run here (the container that has /root/reference); the .desc travels with the repo.  Usage: make_proto_descriptors.py [proto_dir]"""
import os
import re
import sys

from google.protobuf import descriptor_pb2 as D

FILES = ["types.proto", "literal.proto", "expr.proto", "partitioning.proto", "operator.proto", "config.proto", "metric.proto"]
SCALARS = {"double": 1, "float": 2, "int64": 3, "uint64": 4, "int32": 5, "fixed64": 6, "fixed32": 7, "bool": 8, "string": 9,
           "bytes": 12, "uint32": 13, "sfixed32": 15, "sfixed64": 16, "sint32": 17, "sint64": 18}
TOKEN = re.compile(r'"(?:[^"\\]|\\.)*"|[A-Za-z_][A-Za-z0-9_.]*|-?\d+|[{}\[\]<>=;,()]')


def tokenize(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return TOKEN.findall(text)


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def next(self):
        v = self.t[self.i]
        self.i += 1
        return v

    def expect(self, v):
        g = self.next()
        assert g == v, f"expected {v!r}, got {g!r} at token {self.i}"

    def skip_statement(self):
        depth = 0
        while True:
            v = self.next()
            if v in "{[":
                depth += 1
            elif v in "}]":
                depth -= 1
            elif v == ";" and depth == 0:
                return

    def skip_options(self):
        if self.peek() == "[":
            while self.next() != "]":
                pass

    def file(self, name):
        f = D.FileDescriptorProto(name=name, syntax="proto3")
        while self.peek() is not None:
            v = self.next()
            if v == "syntax":
                self.skip_statement()
            elif v == "package":
                f.package = self.next()
                self.expect(";")
            elif v == "import":
                if self.peek() in ("public", "weak"):
                    self.next()
                f.dependency.append(self.next().strip('"'))
                self.expect(";")
            elif v == "option":
                self.skip_statement()
            elif v == "message":
                self.message(f.message_type.add())
            elif v == "enum":
                self.enum(f.enum_type.add())
            elif v == ";":
                pass
            else:
                raise AssertionError(f"unexpected top-level token {v!r}")
        return f

    def enum(self, e):
        e.name = self.next()
        self.expect("{")
        while self.peek() != "}":
            v = self.next()
            if v in ("option", "reserved"):
                self.skip_statement()
                continue
            if v == ";":
                continue
            self.expect("=")
            num = int(self.next())
            self.skip_options()
            self.expect(";")
            e.value.add(name=v, number=num)
        self.expect("}")

    def field(self, m, first, oneof_index=None):
        label, proto3_optional = D.FieldDescriptorProto.LABEL_OPTIONAL, False
        if first == "repeated":
            label, first = D.FieldDescriptorProto.LABEL_REPEATED, self.next()
        elif first == "optional":
            proto3_optional, first = True, self.next()
        if first == "map":
            self.expect("<")
            kt = self.next()
            self.expect(",")
            vt = self.next()
            self.expect(">")
            name = self.next()
            self.expect("=")
            num = int(self.next())
            self.skip_options()
            self.expect(";")
            entry = m.nested_type.add(name="".join(p.capitalize() for p in name.split("_")) + "Entry")
            entry.options.map_entry = True
            self.set_type(entry.field.add(name="key", number=1, label=D.FieldDescriptorProto.LABEL_OPTIONAL), kt)
            self.set_type(entry.field.add(name="value", number=2, label=D.FieldDescriptorProto.LABEL_OPTIONAL), vt)
            fd = m.field.add(name=name, number=num, label=D.FieldDescriptorProto.LABEL_REPEATED)
            fd.type_name = entry.name
            return
        name = self.next()
        self.expect("=")
        num = int(self.next())
        self.skip_options()
        self.expect(";")
        fd = m.field.add(name=name, number=num, label=label)
        self.set_type(fd, first)
        if oneof_index is not None:
            fd.oneof_index = oneof_index
        if proto3_optional:
            fd.proto3_optional = True
            fd.oneof_index = len(m.oneof_decl)
            m.oneof_decl.add(name="_" + name)

    @staticmethod
    def set_type(fd, tname):
        if tname in SCALARS:
            fd.type = SCALARS[tname]
        else:
            fd.type_name = tname          # resolved to a full name (and message vs enum) later

    def message(self, m):
        m.name = self.next()
        self.expect("{")
        synthetic = []
        while self.peek() != "}":
            v = self.next()
            if v == "message":
                self.message(m.nested_type.add())
            elif v == "enum":
                self.enum(m.enum_type.add())
            elif v in ("option", "reserved", "extensions"):
                self.skip_statement()
            elif v == "oneof":
                idx = len(m.oneof_decl)
                m.oneof_decl.add(name=self.next())
                self.expect("{")
                while self.peek() != "}":
                    w = self.next()
                    if w == "option":
                        self.skip_statement()
                    else:
                        self.field(m, w, idx)
                self.expect("}")
            elif v == ";":
                pass
            else:
                self.field(m, v)
        self.expect("}")
        # protoc orders the synthetic oneofs of proto3 `optional` fields after the real ones
        real = [o for o in m.oneof_decl if not o.name.startswith("_")]
        syn = [o for o in m.oneof_decl if o.name.startswith("_")]
        if syn and real:
            order = {o.name: i for i, o in enumerate(real + syn)}
            old = [o.name for o in m.oneof_decl]
            for fd in m.field:
                if fd.HasField("oneof_index"):
                    fd.oneof_index = order[old[fd.oneof_index]]
            names = [o.name for o in real + syn]
            del m.oneof_decl[:]
            for nme in names:
                m.oneof_decl.add(name=nme)


def resolve(files):
    """type_name → '.full.name' and TYPE_MESSAGE / TYPE_ENUM, with protobuf's innermost-scope-first lookup."""
    kinds = {}

    def walk(prefix, msgs, enums):
        for e in enums:
            kinds[prefix + e.name] = D.FieldDescriptorProto.TYPE_ENUM
        for m in msgs:
            kinds[prefix + m.name] = D.FieldDescriptorProto.TYPE_MESSAGE
            walk(prefix + m.name + ".", m.nested_type, m.enum_type)
    for f in files:
        walk(f.package + "." if f.package else "", f.message_type, f.enum_type)

    def fix(scope, m):
        here = scope + m.name
        for fd in m.field:
            if fd.type_name:
                parts = here.split(".")
                full = None
                for k in range(len(parts), -1, -1):
                    cand = ".".join(parts[:k] + [fd.type_name])
                    if cand in kinds:
                        full = cand
                        break
                assert full, f"cannot resolve {fd.type_name} in {here}"
                fd.type_name = "." + full
                fd.type = kinds[full]
        for nm in m.nested_type:
            fix(here + ".", nm)
    for f in files:
        for m in f.message_type:
            fix(f.package + "." if f.package else "", m)


def build(proto_dir):
    files = []
    for name in FILES:
        with open(os.path.join(proto_dir, name)) as fh:
            files.append(Parser(tokenize(fh.read())).file(name))
    resolve(files)
    fds = D.FileDescriptorSet()
    fds.file.extend(files)
    return fds


def main():
    proto_dir = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/native/proto/src/proto"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comet_protos.desc")
    data = build(proto_dir).SerializeToString(deterministic=True)
    with open(out, "wb") as fh:
        fh.write(data)
    print(f"wrote {out}: {len(data)} bytes")


if __name__ == "__main__":
    main()
