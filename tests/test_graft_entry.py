"""The driver's "does it build" check, run as a test: __graft_entry__.build() compiles the HIP library for gfx950, the
oracle and (where the reference checkout exists) the reference-bound binaries, then loads the library and compares its
ABI version with include/katamx.h. No GPU needed; incremental, so it costs seconds after the session fixture built."""
import __graft_entry__ as entry


def test_build_entry_point_succeeds():
    entry.build()


def test_smoke_entry_point_exists():
    assert callable(entry.smoke)
