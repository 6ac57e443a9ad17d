"""CPU-side tests: the C ABI library loads and exports every declared symbol, and the host
mirror's string/container logic (member ranking, HashMap iteration order) agrees with the
oracle's independent model.  No compute calls -- there is no GPU here."""
import ctypes
import os
import random
import re

import pytest

from kafka_lag_based_assignor_amd import _native
from oracle import java_collections as jc
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host():
    from kafka_lag_based_assignor_amd import _host
    return _host


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lagassign.h")).read()
    declared = set(re.findall(r"\b(la_[a-z_]+)\s*\(", header))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.la_version() >= 100


def test_device_batch_struct_matches_header_layout():
    # 4 x int32 + 4 x int64 + 13 pointers + 2 x int64 (the LA_FLAG_BOUNDS hints) + pointer + 2 x int32 (LA_FLAG_WIRE_OUT)
    assert ctypes.sizeof(_native.DeviceBatch) == 16 + 32 + 13 * 8 + 16 + 8 + 8
    # ... and it is the C compiler's layout of the header's struct (offsets of the fields added since ABI 0.3.0)
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "lagassign.h"\nint main(void){printf("%zu %zu %zu %zu %zu",sizeof(la_device_batch),' \
          'offsetof(la_device_batch,max_lag_hint),offsetof(la_device_batch,d_out_wire),offsetof(la_device_batch,wire_id_bits),sizeof(la_call_hints));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    B = _native.DeviceBatch
    assert got == [ctypes.sizeof(B), B.max_lag_hint.offset, B.d_out_wire.offset, B.wire_id_bits.offset, ctypes.sizeof(_native.CallHints)]


def test_binding_constants_match_the_header():
    import re
    header = open(os.path.join(ROOT, "include", "lagassign.h")).read()
    defines = {m.group(1): int(m.group(2), 0)
               for m in re.finditer(r"^#define\s+(LA_[A-Z0-9_]+)\s+\(?(-?(?:0x[0-9a-fA-F]+|\d+))\)?", header, re.M)}
    assert {"LA_OK", "LA_EINVAL", "LA_RESET_LATEST", "LA_ALGO_ARGMIN", "LA_FLAG_RAGGED"} <= set(defines)
    checked = 0
    for name, value in defines.items():
        if hasattr(_native, name):
            assert getattr(_native, name) == value, name
            checked += 1
    assert checked >= 12
    # the device batch: field order and types as declared in the header
    body = header[header.index("typedef struct la_device_batch {"):header.index("} la_device_batch;")]
    fields = re.findall(r"^\s*(?:const\s+)?(int32_t|int64_t|void)\s*(\*?)\s*(\w+);", body, re.M)
    declared = [(n, ("p" if star else t)) for t, star, n in fields]
    bound = [(n, ("p" if (isinstance(t, type) and issubclass(t, (ctypes._Pointer, ctypes.c_void_p))) else
                  {ctypes.c_int32: "int32_t", ctypes.c_int64: "int64_t"}[t])) for n, t in _native.DeviceBatch._fields_]
    assert declared == bound


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(_native.LagAssignError) as ei:
        _native.Context(0)
    assert ei.value.code in (_native.LA_ENODEV, _native.LA_EHIP)


def _rand_ids(rng, n):
    alphabet = ["a", "b", "-", "0", "1", "2", "9", "Z", "é", "～", "\U0001F600", "consumer-"]
    out = set()
    while len(out) < n:
        out.add("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 6))))
    return sorted(out, key=lambda s: rng.random())


def test_java_string_primitives_match_oracle():
    h = _host()
    rng = random.Random(7)
    ids = _rand_ids(rng, 200) + ["hello", "", "topic1", "topic2"]
    for s in ids:
        assert h.java_string_hash(s) == jc.java_string_hash(s) == oracle.java_string_hash(s)
    assert h.java_string_hash("hello") == 99162322
    for _ in range(2000):
        a, b = rng.choice(ids), rng.choice(ids)
        sign = lambda x: (x > 0) - (x < 0)
        assert sign(h.java_string_compare(a, b)) == sign(jc.java_string_compare(a, b)) \
            == sign(oracle.java_string_compare(a, b))


def test_rank_members_is_string_compare_order():
    h = _host()
    ids = ["consumer-%d" % i for i in range(25)] + ["\U0001F600", "～", "Consumer-1"]
    ranks = h.rank_members(ids)
    by_rank = [m for _, m in sorted(zip(ranks, ids))]
    import functools
    assert by_rank == sorted(ids, key=functools.cmp_to_key(jc.java_string_compare))
    assert by_rank.index("consumer-10") < by_rank.index("consumer-2")


def test_reset_mode_parsing():
    h = _host()
    for s, want in [("latest", True), ("LATEST", True), ("LaTeSt", True), ("lateſt", True),
                    ("none", False), ("earliest", False), ("", False), ("latest ", False)]:
        assert h.equals_ignore_case_latest(s) is want


def test_consumers_per_topic_order_matches_oracle_hashmap_model():
    h = _host()
    rng = random.Random(3)
    for trial in range(40):
        topics = ["t%d" % i for i in range(rng.randint(1, 120))] + ["topic1", "topic2"]
        members = ["m%d" % i for i in range(rng.randint(1, 12))]
        subs = [(m, rng.sample(topics, rng.randint(0, len(topics)))) for m in members]
        model = jc.JavaHashMap()
        for m, ts in subs:
            for t in ts:
                model.compute_if_absent(t, list)
        assert h.consumers_per_topic_order(subs) == list(model.keys())


def test_hashmap_put_order_matches_oracle_model():
    h = _host()
    rng = random.Random(5)
    for n in (1, 5, 12, 13, 30, 100, 500):
        keys = _rand_ids(rng, n)
        for cap in (None, 0, 1, 3, n, 2 * n + 1):              # new HashMap<>() and new HashMap<>(cap), Main.java:216
            model = jc.JavaHashMap(cap)
            for k in keys:
                model.put(k, None)
            order, exact = h.hashmap_put_order(keys, cap)
            assert order == list(model.keys()) and exact


def test_hashmap_capacity_constructor_known_orders():
    """`new HashMap<>(3)` is a 4-slot table (tableSizeFor), not 16 slots -- m0, m1, m2 iterate m1, m2, m0
    there and m0, m1, m2 in the default table ("m0".hashCode() = 3427, 3428, 3429: buckets 3, 0, 1 of 4)."""
    h = _host()
    assert jc.java_string_hash("m0") == 3427
    assert h.hashmap_put_order(["m0", "m1", "m2"], 3)[0] == ["m1", "m2", "m0"]
    assert h.hashmap_put_order(["m0", "m1", "m2"])[0] == ["m0", "m1", "m2"]
    m = jc.JavaHashMap(3)
    for k in ("m0", "m1", "m2"):
        m.put(k, 0)
    assert list(m.keys()) == ["m1", "m2", "m0"]
    # thresholds of small tables: 0.75 * capacity truncated (a 2-slot table resizes at its 2nd entry, not its 3rd)
    m = jc.JavaHashMap(2)
    m.put("a", 0)
    assert len(m._table) == 2 and m._threshold == 1
    m.put("b", 0)
    assert len(m._table) == 4 and m._threshold == 3


def _colliding_keys(n, slots=64):
    """n distinct strings whose spread hash lands in bucket 0 of a `slots`-slot table (and of every smaller one)."""
    out, i = [], 0
    while len(out) < n:
        k = "topic-%d" % i
        h = jc.java_string_hash(k) & 0xFFFFFFFF
        if ((h ^ (h >> 16)) & (4 * slots - 1)) == 0:
            out.append(k)
        i += 1
    return out


def test_order_exact_flag_trips_on_a_tree_bin():
    """nine keys in one bucket of a >= 64-slot table is where a real HashMap builds a tree bin.
    Both container models now restate TreeNode's list handling (treeify / putTreeVal / moveRootToFront / split) and give ONE
    definite order; the flag still says that the order came through a tree bin (unverified against a JVM: none here)."""
    h = _host()
    keys = _colliding_keys(9)
    def bucket0(k):
        x = jc.java_string_hash(k) & 0xFFFFFFFF
        return ((x ^ (x >> 16)) & 15) == 0           # bucket 0 of every table size
    # grows the table to 128 slots before the collisions pile up, and stays out of their bucket
    filler = [k for k in ("f%d" % i for i in range(200)) if not bucket0(k)][:60]
    order, exact = h.hashmap_put_order(filler + keys[:8])
    assert exact                                              # eight in a bucket: still a plain chain
    order, exact = h.hashmap_put_order(filler + keys)
    assert not exact
    m = jc.JavaHashMap()
    for k in filler + keys:
        m.put(k, None)
    assert m.treeified and list(m.keys()) == order            # the oracle's restatement walks the same order
    # the root moved to the front: the bucket's keys no longer stand in insertion order
    assert [k for k in order if k in keys] != keys and sorted(k for k in order if k in keys) == sorted(keys)


def _keys_in_bucket(n, mask, want, start=0):
    out, i = [], start
    while len(out) < n:
        k = "t%d" % i
        x = jc.java_string_hash(k) & 0xFFFFFFFF
        if ((x ^ (x >> 16)) & mask) == want:
            out.append(k)
        i += 1
    return out


@pytest.mark.parametrize("case", ["grow_in_place", "split_untreeify", "split_retreeify", "many", "compute_if_absent"])
def test_tree_bin_orders_agree_between_the_two_container_models(case):
    """The C++ host's JavaHashMapOrder and the oracle's JavaHashMap are independent restatements of OpenJDK 8's HashMap, tree
    bins included; on keys chosen to collide they must walk the same order -- through treeification, later insertions into the
    tree (putTreeVal: behind the tree parent, root to the front), table growth that keeps a tree whole ("already treeified"),
    splits it into a tree and a plain chain (untreeify at <= 6) or into two trees (treeified again)."""
    h = _host()
    filler = [k for k in ("f%d" % i for i in range(400)) if (((jc.java_string_hash(k) & 0xFFFFFFFF) ^ ((jc.java_string_hash(k) & 0xFFFFFFFF) >> 16)) & 63) != 5]
    if case == "grow_in_place":            # 14 keys that collide in every table up to 1024 slots; fillers grow the table around the tree
        keys = filler[:60] + _keys_in_bucket(14, 1023, 5) + filler[60:300]
    elif case == "split_untreeify":        # 9 + 3 keys in bucket 5 of a 64-slot table; at 128 slots three of them leave: 9 stay a tree? no: 12 -> 9 / 3
        a = _keys_in_bucket(9, 127, 5)
        b = _keys_in_bucket(3, 127, 5 + 64)
        keys = filler[:40] + a[:5] + b + a[5:] + filler[40:200]
    elif case == "split_retreeify":        # 10 + 9 keys: both halves stay trees after the split
        a = _keys_in_bucket(10, 127, 5)
        b = _keys_in_bucket(9, 127, 5 + 64)
        keys = filler[:40] + [x for pair in zip(a, b) for x in pair] + a[9:] + filler[40:200]
    elif case == "many":
        keys = filler[:50] + _keys_in_bucket(40, 255, 5) + filler[50:120] + _keys_in_bucket(25, 511, 5 + 256, start=50000)
    else:
        keys = filler[:60] + _keys_in_bucket(20, 511, 5) + filler[60:250]
    if case == "compute_if_absent":
        order, exact = h.hashmap_compute_if_absent_order(keys)
        m = jc.JavaHashMap()
        for k in keys:
            m.compute_if_absent(k, lambda: None)
    else:
        order, exact = h.hashmap_put_order(keys)
        m = jc.JavaHashMap()
        for k in keys:
            m.put(k, None)
    assert not exact and m.treeified
    assert list(m.keys()) == order and sorted(order) == sorted(keys)


def test_configure_requires_group_id():                   # Main.java:107-113
    from kafka_lag_based_assignor_amd import LagBasedPartitionAssignor
    a = LagBasedPartitionAssignor()
    with pytest.raises(ValueError):
        a.configure({"auto.offset.reset": "earliest"})
    a.configure({"group.id": "g1", "auto.offset.reset": "earliest", "enable.auto.commit": "true"})
    props = a.metadata_consumer_props()                   # Main.java:116-120
    assert props["enable.auto.commit"] == "false" and props["client.id"] == "g1.assignor"
    assert a.name() == "lag"                              # Main.java:132-135


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    pkg = os.path.join(ROOT, "kafka_lag_based_assignor_amd")
    exe = os.path.join(str(tmp_path), "assign_example")
    subprocess.check_call([cc, "-Wall", "-Werror", "-std=c99", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "assign_example.c"), "-L" + pkg, "-llagassign",
                           "-Wl,-rpath," + pkg, "-o", exe])
    return exe


def test_c_example_compiles_as_plain_c(tmp_path):
    # include/lagassign.h is a C header (C99, no C++), and the example links against the library as a maintainer would
    assert os.path.exists(_build_c_example(tmp_path))


@pytest.mark.gpu
def test_c_example_reproduces_the_readme_example(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C1 (total lag 110000): t0p2 t0p1" in out.stdout


# ---- the Java host (java/): source that cannot be compiled in this image (no JDK) -- what CAN be checked here --------
JAVA_DIR = os.path.join(ROOT, "java")


def test_jni_shim_typechecks_against_the_stub_header():
    """java/jni/lagassign_jni.c against tests/jni_stub/jni.h (JNI-spec signatures, declarations only) and the real
    include/lagassign.h: argument counts and pointer types of every C-ABI call in the shim are checked by the compiler."""
    import shutil
    import subprocess
    if shutil.which("make") is None or (shutil.which("cc") or shutil.which("gcc")) is None:
        pytest.skip("no make / C compiler")
    subprocess.check_call(["make", "-s", "-C", os.path.join(JAVA_DIR, "jni"), "check", "CFLAGS=-O0 -Wall -Wextra -Werror -std=c99 -fPIC"])


def test_java_natives_and_shim_agree():
    """Every `static native` of LagAssignNative.java has its Java_..._<name> function in the shim with the same number
    of parameters, and the shim defines nothing else."""
    import re
    src = open(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/LagAssignNative.java")).read()
    shim = open(os.path.join(JAVA_DIR, "jni/lagassign_jni.c")).read()
    natives = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
               for m in re.finditer(r"static native [\w\[\]]+\s+(\w+)\(([^)]*)\)", src)}
    assert len(natives) >= 10
    shims = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()]) - 2      # minus JNIEnv*, jclass
             for m in re.finditer(r"Java_com_github_grantneale_kafka_gpu_LagAssignNative_(\w+)\(([^)]*)\)", shim)}
    assert natives == shims
    # and the host class only calls natives that exist
    host = open(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/GpuLagBasedPartitionAssignor.java")).read()
    used = set(re.findall(r"LagAssignNative\.(\w+)\(", host))
    assert used and used <= set(natives)


def test_java_build_files_pin_the_reference_versions():
    pom = open(os.path.join(JAVA_DIR, "pom.xml")).read()
    for artifact, version in (("kafka-clients", "2.5.0"), ("slf4j-api", "1.7.30"), ("junit", "4.12"),
                              ("hamcrest-all", "1.3"), ("guava", "21.0")):      # reference pom.xml:86-130
        i = pom.index("<artifactId>%s</artifactId>" % artifact)
        assert "<version>%s</version>" % version in pom[i:i + 200], artifact
    assert os.access(os.path.join(JAVA_DIR, "run_reference_tests.sh"), os.X_OK)
    assert os.path.exists(os.path.join(JAVA_DIR, "src/adapter/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java"))
    # the host logs what the reference logs (LagBasedPartitionAssignor.java:122-128, :268-275, :279-306, :359)
    host = open(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/GpuLagBasedPartitionAssignor.java")).read()
    for text in ("Configured LagBasedPartitionAssignor with values:", "Assignment for {}:\\n{}",
                 "Skipping assignment for topic {} since no metadata is available",
                 "Assigned partition {}-{} to consumer {}.  partition_lag={}, consumer_current_total_lag={}",
                 "\\t%s (total_lag=%d)\\n"):
        assert text in host, text
    assert "IdentityHashMap" not in host and ".stream()" not in host            # the leak, the per-topic streams


def _log_formats(java_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract_log_formats", os.path.join(ROOT, "tools", "extract_log_formats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.formats(open(java_path, encoding="utf-8").read())


def test_java_host_logs_the_reference_messages_byte_for_byte():
    """Every LOGGER.* format string of the reference (Main.java:122-128, :268-275, :299-303, :359; extracted into
    tests/golden/reference_log_formats.json by tools/extract_log_formats.py) appears in the Java host with the same
    level and the same bytes; where the reference checkout is present, the fixture is checked against it too."""
    import json
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_log_formats.json")))["formats"]
    assert [(f["level"], f["line"]) for f in fixture] == [("debug", 122), ("trace", 268), ("debug", 300), ("warn", 359)]
    host = _log_formats(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/GpuLagBasedPartitionAssignor.java"))
    have = {(f["level"], f["format"]) for f in host}
    for f in fixture:
        assert (f["level"], f["format"]) in have, f
    ref = "/root/reference/src/main/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java"
    if os.path.exists(ref):
        assert [(f["level"], f["format"], f["line"]) for f in _log_formats(ref)] == \
               [(f["level"], f["format"], f["line"]) for f in fixture]


def test_java_host_hardening():
    host = open(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/GpuLagBasedPartitionAssignor.java")).read()
    # only native failures are delegated to the fallback class, never what the side KafkaConsumer throws
    assert "catch (NativeAssignException | LinkageError" in host
    assert "catch (IllegalStateException | LinkageError" not in host
    assert "class NativeAssignException extends IllegalStateException" in host
    # no silent truncation above 2 GiB: ByteBuffer capacities are ints
    assert "(int) Math.min(Integer.MAX_VALUE" not in host and "wantBytes > Integer.MAX_VALUE" in host
    shim = open(os.path.join(JAVA_DIR, "jni", "lagassign_jni.c")).read()
    body = shim[shim.index("LagAssignNative_hostAlloc"):shim.index("LagAssignNative_hostFree")]
    assert "la_host_free" in body                              # the pinned block is not leaked when NewDirectByteBuffer fails
    # the adapter must not hide the inherited PUBLIC static with a package-private one (JLS 8.4.8.3: would not compile)
    adapter = _strip_java(open(os.path.join(JAVA_DIR, "src/adapter/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java")).read())
    assert "computePartitionLag" not in adapter
    assert re.search(r"public\s+static\s+(synchronized\s+)?long\s+computePartitionLag", host)
    # "skipped" must not read as "passed"
    script = open(os.path.join(JAVA_DIR, "run_reference_tests.sh")).read()
    assert script.count("exit 3") >= 3 and "exit 0" not in script


@pytest.mark.gpu
def test_reference_junit_class_runs_against_the_java_host():
    """The reference's own LagBasedPartitionAssignorTest.java, unchanged, on the GPU path -- where a JDK and the jars
    exist (not in the image this repository is developed in: skipped there)."""
    import shutil
    import subprocess
    if shutil.which("javac") is None:
        pytest.skip("no JDK on this machine (java/run_reference_tests.sh is ready for one)")
    ref = os.environ.get("LA_REFERENCE_DIR", "/root/reference")
    r = subprocess.run([os.path.join(JAVA_DIR, "run_reference_tests.sh"), ref], capture_output=True, text=True, timeout=1800)
    if r.returncode == 3:
        pytest.skip("prerequisites missing: " + r.stderr.strip())
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]


def _strip_java(src):
    """Java source with comments, string and char literals blanked (delimiters kept out)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            assert j >= 0, "unterminated comment"
            i = j + 2
        elif c in "\"'":
            j = i + 1
            while src[j] != c:
                j += 2 if src[j] == "\\" else 1
                assert j < n and src[j - 1] != "\n", "unterminated literal near offset %d" % i
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_java_sources_are_lexically_sane():
    """No compiler here, so at least: delimiters balance, every file declares the package its path says, every
    import is used, no statement-level typo classes like a stray ';;' after a brace or an unclosed generic."""
    import re
    files = []
    for root, _, names in os.walk(os.path.join(JAVA_DIR, "src")):
        files += [os.path.join(root, n) for n in names if n.endswith(".java")]
    assert len(files) >= 3
    pairs = {")": "(", "]": "[", "}": "{"}
    for path in files:
        code = _strip_java(open(path).read())
        stack = []
        for ch in code:
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], "%s: unbalanced %r" % (path, ch)
        assert not stack, "%s: unclosed %r" % (path, stack[-1])
        pkg = re.search(r"^\s*package\s+([\w.]+)\s*;", code, re.M).group(1)
        assert path.replace(os.sep, ".").endswith(pkg + "." + os.path.basename(path)), path
        for imp in re.findall(r"^\s*import\s+(?:static\s+)?[\w.]+\.(\w+)\s*;", code, re.M):
            body = re.sub(r"^\s*import\s+.*$", "", code, flags=re.M)
            assert re.search(r"\b%s\b" % imp, body), "%s: unused import %s" % (os.path.basename(path), imp)


def test_bench_workloads_are_the_fixture_generator():
    """bench.py times the vectors of synth.config (it used torch's RNG); its sort-phase topic carries a
    permutation of 0..n-1 as partition ids, also for sizes that are not a power of two."""
    import argparse
    import numpy as np
    import bench
    from kafka_lag_based_assignor_amd import synth
    ns = argparse.Namespace(workload="cfg3", topics=None, partitions=None, consumers=None, dist=None)
    w, name, dist = bench.make_workload(ns)
    ref = synth.config("cfg3")
    assert name == "cfg3" and dist == "Zipf(1.1)"
    for k in ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank"):
        np.testing.assert_array_equal(getattr(w, k), getattr(ref, k))
    ns = argparse.Namespace(workload="target", topics=50, partitions=40, consumers=7, dist="uniform40")
    w, name, _ = bench.make_workload(ns)
    assert name == "custom" and (w.n_topics, w.max_partitions, w.max_consumers) == (50, 40, 7)
    for n in (1, 1000, 4096, 100003):
        s = bench.sort_phase_workload(n)
        assert s.n_topics == 1 and s.cons_rank.size == 0 and s.n_partitions == n
        np.testing.assert_array_equal(np.sort(s.partition_id), np.arange(n, dtype=np.int32))
        assert s.lag.min() >= 0 and s.lag.max() < (1 << 40)
        if n > 1000:
            assert np.any(np.diff(s.partition_id) < 0)                 # not already in id order: the id passes run


# ---- round 5 ----------------------------------------------------------------------------------------------------------
def test_java_host_hints_bounds_and_falls_back_to_the_reference_by_default():
    """VERDICT r4 next #1 / #8, lexically (no JDK here): the marshalling loop tracks the largest end offset and partition id
    and hands them over with hintNextCallBounds before the assign call, only when the library has it (ABI 0.4.0) and nothing
    was negative; with lag.assignor.fallback.class unset the reference class is looked up BY NAME (nothing linked) and takes a
    failed rebalance over -- never the test adapter, which carries that name and is this host."""
    import re
    host = open(os.path.join(JAVA_DIR, "src/main/java/com/github/grantneale/kafka/gpu/GpuLagBasedPartitionAssignor.java")).read()
    loop = host[host.index("long maxEnd = 0;"):host.index("partOff.put(nTopics, cursor);")]
    assert "maxEnd = Math.max(maxEnd, e);" in loop and "maxPartition = Math.max(maxPartition, tp.partition());" in loop
    assert "anyNegative |= e < 0 || tp.partition() < 0;" in loop and "anyNegative |= b < 0;" in loop
    i_hint, i_call = host.index("LagAssignNative.hintNextCallBounds(engine.ctx, maxEnd, maxPartition)"), host.index("LagAssignNative.assignBatchGroupedSparse(")
    assert i_hint < i_call and "if (!anyNegative && n > 0 && engine.hasHints)" in host[i_hint - 900:i_hint]
    assert "hasHints = version >= 400;" in host
    # the dense begin array is marshalled only for the trace path that reads it
    assert re.search(r"if \(trace\) \{\s*beginOff\.put\(cursor, b\);", loop)
    # default fall-back: by name, reflection only, not the adapter
    assert 'DEFAULT_FALLBACK_CLASS = "com.github.grantneale.kafka.LagBasedPartitionAssignor"' in host
    fb = host[host.index("private ConsumerPartitionAssignor fallback()"):]
    assert "Class.forName(DEFAULT_FALLBACK_CLASS, false," in fb and "GpuLagBasedPartitionAssignor.class.isAssignableFrom(found)" in fb
    assert "import com.github.grantneale.kafka.LagBasedPartitionAssignor" not in host


def test_library_exports_the_round5_and_round6_symbols_and_version():
    from kafka_lag_based_assignor_amd import _native as N
    lib = N.load()
    assert lib.la_version() == 500
    header = open(os.path.join(ROOT, "include", "lagassign.h")).read()
    assert "#define LA_VERSION 500" in header
    for sym in ("la_hint_next_call", "la_last_launches", "la_last_phase_times_sized", "la_wake"):
        assert sym in header and getattr(lib, sym)
    # la_call_hints of the ctypes binding is the header's struct: 4 + 4 + 8 + 8 bytes
    import ctypes
    assert ctypes.sizeof(N.CallHints) == 24


def test_offset_bounds_is_what_a_marshaller_may_promise():
    from kafka_lag_based_assignor_amd import _native as N
    import numpy as np
    end = np.array([5, 100, 7], np.int64)
    assert N.offset_bounds(np.zeros(3, np.int64), end, np.array([1, -1, 3], np.int64), np.array([2, 0, 9], np.int32)) == (100, 9)
    assert N.offset_bounds(np.array([0, -4, 0], np.int64), end, end, np.array([2, 0, 9], np.int32)) is None      # a negative begin
    assert N.offset_bounds(None, np.array([5, -1], np.int64), np.zeros(2, np.int64), np.array([0, 1], np.int32)) is None
    assert N.offset_bounds(None, end, end, np.array([0, -1, 1], np.int32)) is None                                   # a negative id
    assert N.offset_bounds(None, None, None, np.array([3, 1], np.int32), lag=np.array([9, 4], np.int64)) == (9, 3)
    assert N.offset_bounds(None, None, None, np.array([3, 1], np.int32), lag=np.array([9, -4], np.int64)) is None


def test_tools_and_bench_scripts_compile():
    """The probes, sessions' helpers and bench.py only ever run on the GPU box: a syntax error in one of them would surface there,
    minutes into a session.  Byte-compile them all here."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        with open(f, "rb") as src:
            compile(src.read(), f, "exec")                         # (raises SyntaxError; nothing is written)


def test_binding_refuses_result_buffers_the_library_cannot_fill():
    """_native._check_out3: caller-owned result buffers must be contiguous arrays of the right type and size (the library writes N / K
    elements at one address each); arrays over foreign memory, as host_alloc builds them, are fine."""
    import numpy as np
    n, k = 12, 5
    p, m, t = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(k, np.int64)
    assert _native._check_out3((p, m, t), n, k) == (p, m, t)
    assert _native._check_out3((p, m, None), n, k)[2] is None
    buf = (ctypes.c_char * (4 * n)).from_address(ctypes.addressof(ctypes.create_string_buffer(4 * n)))
    foreign = np.frombuffer(buf, dtype=np.int32, count=n)
    assert _native._check_out3((foreign, m, t), n, k)[0] is foreign
    assert _native._addr(foreign) == foreign.ctypes.data and _native._addr(p) == p.ctypes.data
    ro = np.zeros(3, np.int64)
    ro.flags.writeable = False
    assert _native._addr(ro) == ro.ctypes.data and _native._addr(np.empty(0, np.int32)) is not None and _native._addr(None) is None
    for bad in ((np.empty(2 * n, np.int32)[::2], m, t), (p.astype(np.int64), m, t), (p, m[:-1], t), (p, m, t.astype(np.int32)),
                (p, m, np.empty(k + 1, np.int64)), (list(p), m, t)):
        with pytest.raises(ValueError):
            _native._check_out3(bad, n, k)
