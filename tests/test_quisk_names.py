"""The module is name-complete: every method of the reference's _quisk table (quisk.c:5164-5266) and every `QS.<name>` that the
reference's Python files use exists in the C extension and in the Python shim, callable; the entries that are off the
receive path return the reference's shapes.  The lists (names and PyArg formats are interface facts) were read from the
reference's text by tests/golden/make_qs_names.py."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NAMES = json.load(open(os.path.join(HERE, "golden", "qs_names.json")))


def _ext():
    sys.path.insert(0, os.path.join(ROOT, "quisk_amd", "ext"))
    import _quisk
    return _quisk


def _shim():
    sys.path.insert(0, ROOT)
    from quisk_amd import _quisk
    return _quisk


@pytest.mark.parametrize("which", ["extension", "shim"])
def test_every_reference_name_is_there(which):
    QS = _ext() if which == "extension" else _shim()
    assert len(NAMES["method_table"]) == 100 and len(NAMES["used_by_python"]) > 80
    missing = [n for n in NAMES["method_table"] + NAMES["used_by_python"] if not hasattr(QS, n)]
    assert missing == []
    assert all(callable(getattr(QS, n)) for n in NAMES["method_table"])
    assert hasattr(QS, "error") and issubclass(QS.error, Exception)


@pytest.mark.parametrize("which", ["extension", "shim"])
def test_entries_off_the_path_keep_the_reference_shapes(which):
    """callable with no engine and the reference's arity ('' formats), returning what the reference's callers unpack"""
    QS = _ext() if which == "extension" else _shim()
    assert QS.sound_devices() == ([], []) and QS.pa_sound_devices() == ([], [])          # quisk.py:1284: two lists
    assert QS.sound_errors() == []                                                        # quisk.py:951
    assert QS.ip_interfaces() == []
    assert QS.get_hermes_TFRC() == (0.0, 0.0, 0.0, 0.0)
    assert bytes(QS.get_hermeslite_response()) == bytes(5) and bytes(QS.hermes_to_pc()) == bytes(20)
    assert QS.get_hermeslite_writepointer() == 0 and QS.freedv_get_version() == -1
    assert QS.freedv_get_snr() == 0.0 and QS.freedv_get_rx_char() == ""
    assert QS.test_1() is None and QS.test_2() is None and QS.test_3() is None
    assert QS.is_key_down() == 0


def test_zero_argument_formats_reject_arguments_in_the_extension():
    """the entries with a '' PyArg format that this module implements parse their arguments like the reference"""
    QS = _ext()
    for n in ("sound_devices", "sound_errors", "get_hermes_adc", "get_hermes_TFRC", "test_1", "freedv_get_snr", "get_tx_filter"):
        assert NAMES["formats"][n] == ""
        with pytest.raises(TypeError):
            getattr(QS, n)(1)
