#!/bin/bash
# Compiles the Java host + JNI shim and runs the REFERENCE'S OWN JUnit class
# (grantneale/kafka-lag-based-assignor src/test/java/.../LagBasedPartitionAssignorTest.java, unchanged) against the
# GPU path, through the adapter in java/src/adapter.  Needs: a JDK 8+, an MI355X, liblagassign.so (built first), and
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

if command -v mvn >/dev/null 2>&1 && [ -z "${LA_JARS:-}" ]; then
  exec mvn -q -f "$HERE/pom.xml" -Preference-tests -Dreference.dir="$REF" -Dnative.dir="$NATIVE" test
fi

# no Maven: plain javac + JUnitCore over a directory of jars
# (kafka-clients-2.5.0, slf4j-api-1.7.30, junit-4.12, hamcrest-all-1.3, guava-21.0 and any slf4j binding)
JARS="${LA_JARS:-}"
[ -n "$JARS" ] && ls "$JARS"/*.jar >/dev/null 2>&1 || { echo "run_reference_tests: no mvn and no LA_JARS directory" >&2; exit 3; }
CP="$(ls "$JARS"/*.jar | tr '\n' ':')"
OUT="$(mktemp -d)"
trap 'rm -rf "$OUT"' EXIT
javac -source 8 -target 8 -nowarn -d "$OUT" -cp "$CP" \
  $(find "$HERE/src/main/java" "$HERE/src/adapter/java" -name '*.java') \
  "$REF/src/test/java/com/github/grantneale/kafka/LagBasedPartitionAssignorTest.java"
LD_LIBRARY_PATH="$NATIVE:/opt/rocm/lib:${LD_LIBRARY_PATH:-}" \
  java -Djava.library.path="$NATIVE" -cp "$OUT:$CP" org.junit.runner.JUnitCore \
  com.github.grantneale.kafka.LagBasedPartitionAssignorTest
