"""The ROS parameter server's content for the node of a roslaunch file, read INDEPENDENTLY of the product's loader -- test infrastructure only.

roslaunch semantics restated for what /root/reference/swarm_loop/launch/*.launch use (roslaunch/xmlloader.py, roslaunch/loader.py): <arg name default|value> with
command-line overrides, $(arg x) substitution ($(find pkg) is left verbatim), if= / unless= on the node, then the node's children in document order:
<rosparam> = yaml.safe_load of the substituted text (PyYAML: the library roslaunch itself calls, YAML 1.1 typing -- `1e-2` is a string), <param name value type>
= roslaunch.loader.convert_value.  Returns [(name, 'I' | 'D' | 'B' | 'S', text)] with later settings of a name replacing earlier ones: the typed table
tests/cpp/params_pin.cpp feeds to the reference's own parameter block."""
from __future__ import annotations

import re
import xml.etree.ElementTree as ET

import yaml


def _convert_value(value: str, type_: str):
    """roslaunch.loader.convert_value"""
    type_ = (type_ or "auto").lower().strip()
    if type_ == "auto":
        try:
            return float(value) if "." in value else int(value)
        except ValueError:
            pass
        if value.lower() in ("true", "false"):
            return value.lower() == "true"
        return value
    if type_ in ("str", "string"):
        return value
    if type_ == "int":
        return int(value)
    if type_ == "double":
        return float(value)
    if type_ in ("bool", "boolean"):
        v = value.lower().strip()
        if v in ("true", "1"):
            return True
        if v in ("false", "0"):
            return False
        raise ValueError(f"{value} is not a bool")
    raise ValueError(f"unknown type {type_}")


def parameter_server(xml_text: str, node_name: str = "swarm_loop", args: dict | None = None):
    args = dict(args or {})
    root = ET.fromstring(xml_text)
    declared: dict[str, str] = {}

    def subst(s: str) -> str:
        def rep(m):
            verb, what = m.group(1), m.group(2).strip()
            if verb == "arg":
                return declared[what]
            return m.group(0)                      # $(find pkg): no ROS here, kept verbatim
        return re.sub(r"\$\((arg|find)\s+([^)]*)\)", rep, s)

    server: dict[str, object] = {}
    found = False
    for el in root:                                # document order
        if el.tag == "arg":
            n = el.get("name")
            if el.get("value") is not None:
                declared[n] = subst(el.get("value"))
            elif n in args:
                declared[n] = args[n]
            else:
                declared[n] = subst(el.get("default"))
        elif el.tag == "node" and el.get("name") == node_name:
            if el.get("if") is not None and not _convert_value(subst(el.get("if")), "bool"):
                continue
            if el.get("unless") is not None and _convert_value(subst(el.get("unless")), "bool"):
                continue
            found = True
            for c in el:
                if c.tag == "rosparam":
                    data = yaml.safe_load(subst(c.text or "")) or {}
                    for k, v in data.items():
                        if v is not None:
                            server[str(k)] = v
                elif c.tag == "param":
                    server[c.get("name")] = _convert_value(subst(c.get("value")), c.get("type"))
    if not found:
        raise ValueError(f"no <node name={node_name!r}>")
    out = []
    for k, v in server.items():
        if isinstance(v, bool):
            out.append((k, "B", "true" if v else "false"))
        elif isinstance(v, int):
            out.append((k, "I", str(v)))
        elif isinstance(v, float):
            out.append((k, "D", repr(v)))
        else:
            out.append((k, "S", str(v)))
    return out
