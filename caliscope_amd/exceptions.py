"""Error types of the BA path (mirror of reference ``src/caliscope/exceptions.py:1-14``)."""


class CalibrationError(Exception):
    """A calibration operation failed; the message says what to do about it."""


class CalibrationWarning(UserWarning):
    """Non-fatal calibration issue."""


class BackendError(RuntimeError):
    """The MI355X backend (libcaliscope_ba.so / HIP device) is missing or reported an error.

    There is deliberately no CPU fallback: a solve either runs on the HIP kernels or raises this.
    """
