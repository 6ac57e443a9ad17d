"""Extracts the slf4j format string of every LOGGER.<level>(...) call of a Java source: the literals up to the first
argument, concatenated.  Used twice: on the reference's LagBasedPartitionAssignor.java to write the committed fixture
tests/golden/reference_log_formats.json (run it in the build container, where /root/reference exists), and by
tests/test_host_cpu.py on the Java host of java/ to check that it logs the reference's four messages byte for byte.

    python tools/extract_log_formats.py /root/reference/src/main/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java \
        > tests/golden/reference_log_formats.json
"""
import json
import re
import sys

CALL = re.compile(r'LOGGER\s*\.\s*(trace|debug|info|warn|error)\s*\(\s*((?:"(?:[^"\\]|\\.)*"\s*(?:\+\s*)?)+)', re.S)
LIT = re.compile(r'"((?:[^"\\]|\\.)*)"')


def formats(java_source: str):
    out = []
    for m in CALL.finditer(java_source):
        out.append({"level": m.group(1), "format": "".join(LIT.findall(m.group(2))),
                    "line": java_source.count("\n", 0, m.start()) + 1})
    return out


if __name__ == "__main__":
    src = open(sys.argv[1], encoding="utf-8").read()
    json.dump({"source": "LagBasedPartitionAssignor.java of grantneale/kafka-lag-based-assignor v2.0.0 (LOGGER call sites)",
               "formats": formats(src)}, sys.stdout, indent=1)
    sys.stdout.write("\n")
