#!/bin/bash
# Compiles the Java host + JNI shim and runs the REFERENCE'S OWN JUnit class
# (grantneale/kafka-lag-based-assignor src/test/java/.../LagBasedPartitionAssignorTest.java, unchanged) against the
# GPU path, through the adapter in java/src/adapter; then times the reference's OWN assign(Map,Map) on the bench vectors and
# prints the bench line's cpu_baseline object with "kind": "reference" (the north star's Java baseline).  Needs: a JDK 8+, an MI355X, liblagassign.so (built first), and
# either Maven with access to the four artifacts of java/pom.xml or a directory of their jars (LA_JARS).
#
#   java/run_reference_tests.sh [REFERENCE_DIR]        # default /root/reference
#
# Exit code 3 = prerequisites missing (no JDK / no jars): nothing was run.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${1:-/root/reference}"
NATIVE="$ROOT/kafka_lag_based_assignor_amd"

command -v javac >/dev/null 2>&1 || { echo "run_reference_tests: no javac on PATH" >&2; exit 3; }
[ -f "$REF/src/test/java/com/github/grantneale/kafka/LagBasedPartitionAssignorTest.java" ] || {
  echo "run_reference_tests: $REF is not a checkout of the reference" >&2; exit 3; }
[ -f "$NATIVE/liblagassign.so" ] || (cd "$ROOT" && python -m kafka_lag_based_assignor_amd.build)
make -C "$HERE/jni" ROOT="$ROOT"

# javac flags: the reference builds with -Xlint:all and failOnWarning (its pom.xml:136-149); so does the host's main source.
# JDK 9+ warns about "-source 8" without a bootstrap class path, which -Werror would turn into a failure: --release 8 there.
JV="$(javac -version 2>&1 | sed -E 's/^javac (1\.)?([0-9]+).*/\2/')"
if [ "${JV:-8}" -ge 9 ] 2>/dev/null; then REL="--release 8"; else REL="-source 8 -target 8"; fi

baseline() {
  # The north star's CPU baseline: the reference's OWN static assign(Map,Map) (Main.java:166), compiled from its own source,
  # single thread, on the first LA_BASELINE_TOPICS topics of the bench workload (tools/export_vectors.py writes them with the
  # oracle's checksum).  Prints one JSON object: {"cpu_baseline": {..., "kind": "reference"}}; exit 1 if the reference's
  # assignment differs from the oracle's.
  local cp="$1" out vec
  out="$(mktemp -d)"; vec="$out/vectors.lav1"
  python "$ROOT/tools/export_vectors.py" --workload "${LA_BASELINE_WORKLOAD:-target}" --topics "${LA_BASELINE_TOPICS:-2000}" --out "$vec" >&2
  javac $REL -nowarn -d "$out" -cp "$cp" \
    "$REF/src/main/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java" \
    "$HERE/src/bench/java/com/github/grantneale/kafka/ReferenceBaseline.java"
  java -Xmx8g -cp "$out:$cp" com.github.grantneale.kafka.ReferenceBaseline "$vec" "${LA_BASELINE_SECONDS:-10}"
  local rc=$?
  rm -rf "$out"
  return $rc
}

if command -v mvn >/dev/null 2>&1 && [ -z "${LA_JARS:-}" ]; then
  mvn -q -f "$HERE/pom.xml" -Preference-tests -Dreference.dir="$REF" -Dnative.dir="$NATIVE" test
  CP="$(mvn -q -f "$HERE/pom.xml" dependency:build-classpath -Dmdep.outputFile=/dev/stdout)"
  baseline "$CP"
  exit $?
fi

# no Maven: plain javac + JUnitCore over a directory of jars
# (kafka-clients-2.5.0, slf4j-api-1.7.30, junit-4.12, hamcrest-all-1.3, guava-21.0 and any slf4j binding)
JARS="${LA_JARS:-}"
[ -n "$JARS" ] && ls "$JARS"/*.jar >/dev/null 2>&1 || { echo "run_reference_tests: no mvn and no LA_JARS directory" >&2; exit 3; }
CP="$(ls "$JARS"/*.jar | tr '\n' ':')"
OUT="$(mktemp -d)"
trap 'rm -rf "$OUT"' EXIT
# the host's main source as strictly as the reference's build compiles its own
javac $REL -Xlint:all -Werror -d "$OUT" -cp "$CP" $(find "$HERE/src/main/java" -name '*.java')
# the adapter and the reference's JUnit class, unchanged
javac $REL -nowarn -d "$OUT" -cp "$OUT:$CP" \
  $(find "$HERE/src/adapter/java" -name '*.java') \
  "$REF/src/test/java/com/github/grantneale/kafka/LagBasedPartitionAssignorTest.java"
LD_LIBRARY_PATH="$NATIVE:/opt/rocm/lib:${LD_LIBRARY_PATH:-}" \
  java -Djava.library.path="$NATIVE" -cp "$OUT:$CP" org.junit.runner.JUnitCore \
  com.github.grantneale.kafka.LagBasedPartitionAssignorTest
baseline "$CP"
