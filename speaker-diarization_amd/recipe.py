"""AKU recipe text format: reader, writer and the Python-2 number formatting the
inter-stage contract depends on.

Reference behaviour restated here (paths relative to the reference tree):
  * line grammar and "echo + skip" on unparsable lines .... spk-change-detection.py:11-28,
    spk-clustering.py:11-28, spk-clustering2.py:12-29
  * writer, `lna` renaming state machine ................... spk-change-detection.py:46-69,
    spk-clustering.py:55-78
  * every time is written with Python-2 ``str(float)`` (12 significant digits) and
    re-read by the next stage.
"""
import re

_AUDIO = re.compile(r'audio=(\S+)')
_LNA = re.compile(r'lna=(\S+)')
# NB: the unescaped '.' is part of the contract (any character between digit runs).
_START = re.compile(r'start-time=(\d+.\d+)')
_END = re.compile(r'end-time=(\d+.\d+)')


def py2_float_str(x):
    """``str(x)`` of a float as CPython 2.7 printed it: ``'%.12g'`` plus a
    trailing ``.0`` when the result would otherwise read as an integer."""
    x = float(x)
    if x != x:
        return 'nan'
    if x in (float('inf'), float('-inf')):
        return 'inf' if x > 0 else '-inf'
    s = '%.12g' % x
    if '.' not in s and 'e' not in s:
        s += '.0'
    return s


def py2_str(x):
    """``str`` for the values the scripts print: floats (incl. numpy floats) the
    py2 way, everything else as py3 does."""
    import numbers
    if isinstance(x, bool):
        return str(x)
    if isinstance(x, numbers.Real) and not isinstance(x, numbers.Integral):
        return py2_float_str(x)
    try:  # numpy scalar floats are numbers.Real; numpy ints are Integral
        import numpy as np
        if isinstance(x, np.floating):
            return py2_float_str(x)
    except ImportError:  # pragma: no cover
        pass
    return str(x)


class RecipeLine(tuple):
    """(audio, lna, start_seconds, end_seconds) — same positional layout as the
    tuples the reference passes around."""
    __slots__ = ()

    def __new__(cls, audio, lna, start, end):
        return tuple.__new__(cls, (audio, lna, float(start), float(end)))

    audio = property(lambda s: s[0])
    lna = property(lambda s: s[1])
    start = property(lambda s: s[2])
    end = property(lambda s: s[3])


def parse_recipe(lines, echo=None):
    """Parse an iterable of recipe lines.  Lines lacking any of the four fields
    are echoed through ``echo`` (two calls, like the two py2 print statements)
    and skipped."""
    out = []
    for line in lines:
        a = _AUDIO.search(line)
        l = _LNA.search(line)
        s = _START.search(line)
        e = _END.search(line)
        if a is None or l is None or s is None or e is None:
            if echo is not None:
                echo('Recipe line without recognizable data:')
                echo(line)
            continue
        out.append(RecipeLine(a.group(1), l.group(1), float(s.group(1)), float(e.group(1))))
    return out


class RecipeWriter(object):
    """Stateful line writer.  Holds the (lna_letter, lna_count) pair that the
    reference keeps in module globals, so one writer == one script run."""

    def __init__(self, outf, rate, segf=None, segpath=None, rename_lna=True):
        self.outf = outf
        self.segf = segf
        self.segpath = segpath
        self.rate = float(rate)
        self.rename = rename_lna
        self.lna_letter = 'a'
        self.lna_count = 0

    def _lna(self, lna):
        if not self.rename:
            return lna
        cut = lna.find('_')           # -1 when absent: prefix = all but last char
        prefix = lna[:cut]
        if prefix == self.lna_letter:
            self.lna_count += 1
        else:
            self.lna_count = 1
            self.lna_letter = prefix
        return lna[:cut + 1] + str(self.lna_count)

    def write(self, recline, start_frames, end_frames, lna_start, speaker):
        """``speaker`` is the full label text (``spk_turn`` / ``speaker_3``)."""
        lna = self._lna(recline[1])
        t0 = py2_float_str(start_frames / self.rate + lna_start)
        t1 = py2_float_str(end_frames / self.rate + lna_start)
        tail = ' lna=' + lna + ' start-time=' + t0 + ' end-time=' + t1 + \
               ' speaker=' + speaker + '\n'
        self.outf.write('audio=' + recline[0] + tail)
        if self.segpath and self.segf is not None:
            self.segf.write('audio=' + recline[0] + ' alignment=' + self.segpath +
                            lna + '.seg' + tail)


def roundtrip_time(frames, rate, lna_start):
    """The value the next stage reads back for a boundary this stage writes:
    format with 12 digits, re-parse."""
    return float(py2_float_str(frames / float(rate) + lna_start))
