"""Diagnostic marks of the launch thread (ALIBY_RUNNER_TRACE=1): (label, host clock) pairs, reported by
aliby_amd.runner.run_positions(stats=...) — where the host is between two device synchronisations."""
import os
import time

MARKS = [] if os.environ.get("ALIBY_RUNNER_TRACE") else None


def mark(label: str) -> None:
    if MARKS is not None:
        MARKS.append((label, time.perf_counter()))


# Hooks run by the launch thread right before it blocks in a long device synchronisation (the segmenter's dynamics wait for
# the whole network): the position-batched runner hands its queued file-writing tasks to the writer threads at that moment,
# so that their Python runs while the launch thread sleeps instead of competing with it for the interpreter lock.
BEFORE_BLOCK: list = []


def about_to_block() -> None:
    for hook in list(BEFORE_BLOCK):
        hook()

