"""Downstream consumers of the output recipe (SURVEY.md §8(f) "next" row 3), host side:

  * aku2ann  -- recipe -> simple tab-separated annotation ......... aku2ann.py:7-38, 41-69
  * DER      -- frame-level (1 ms) speaker-error rate of a proposed recipe against a
                baseline, greedy one-to-one-by-count label matching  clus-performance.py:9-108

  * aku2elan -- recipe -> ELAN .eaf (EAF 2.7 XML) ................... aku2elan.py:10-99
                PARITY UNPINNED: the reference writes through lxml, which is not installed
                here, so no reference output could be generated; the document structure,
                attribute values and the two-space pretty print follow the source, the
                attribute order is the one the source spells (Python 2 / lxml order is a
                dict-hash artefact), and DATE is the wall clock as in the reference.
"""
import argparse
import re
import sys

import numpy as np

from .recipe import py2_str

_AUDIO = re.compile(r'audio=(\S+)')
_LNA = re.compile(r'lna=(\S+)')
_START = re.compile(r'start-time=(\d+.\d+)')      # the unescaped '.' is the reference's
_END = re.compile(r'end-time=(\d+.\d+)')
_SPEAKER = re.compile(r'speaker=(\S+)')


def _say_to(out):
    def say(*items):
        out.write(' '.join(py2_str(x) for x in items) + '\n')
    return say


# --------------------------------------------------------------------------- aku2ann
def parse_recipe_ann(lines, say):
    """(audio, lna, start, end, speaker); lines missing audio / lna / a time are reported
    (the line rides on the same print, aku2ann.py:27) and skipped."""
    out = []
    for line in lines:
        a, l, s, e = _AUDIO.search(line), _LNA.search(line), _START.search(line), _END.search(line)
        if not (a and l and s and e):
            say('Recipe line without recognizable data:', line)
            continue
        sp = _SPEAKER.search(line)
        out.append((a.group(1), l.group(1), float(s.group(1)), float(e.group(1)), sp.group(1) if sp else ''))
    return out


def write_ann(recipe, outf):
    audio = ''
    for line in recipe:
        if audio != line[0]:
            audio = line[0]
            outf.write('# ' + audio + '\n')
        outf.write(py2_str(line[2]) + '\t' + py2_str(line[3]) + '\t' + line[4] + '\n')


def main_aku2ann(argv=None, stdout=None):
    out = stdout or sys.stdout
    say = _say_to(out)
    p = argparse.ArgumentParser(description='Converts an AKU recipe to simple annotation format.')
    p.add_argument('recfile', type=str, help='Specifies the input recipe file')
    p.add_argument('-o', dest='outfile', type=str, default=None, help='Specifies an output file, default stdout.')
    args = p.parse_args(argv)
    say('Reading recipe from:', args.recfile)
    with open(args.recfile, 'r') as f:
        recipe = parse_recipe_ann(f, say)
    if args.outfile is not None:
        say('Writing output to:', args.outfile)
        with open(args.outfile, 'w') as outf:
            write_ann(recipe, outf)
    else:
        say('Writing output to: stdout')
        write_ann(recipe, out)
    return None


# --------------------------------------------------------------------------- aku2elan
def date_iso(now=None, utcnow=None):
    """Local time in ISO format with the UTC offset, as ELAN expects it (aku2elan.py:10-17)."""
    from datetime import datetime
    dtnow = now or datetime.now()
    dtutcnow = utcnow or datetime.utcnow()
    delta = dtnow - dtutcnow
    hh, mm = divmod((delta.days * 24 * 60 * 60 + delta.seconds + 30) // 60, 60)
    return '%s%+02d:%02d' % (dtnow.isoformat(), hh, mm)


def _xml_attr(v):
    return v.replace('&', '&amp;').replace('<', '&lt;').replace('"', '&quot;')


def _xml_text(v):
    return v.replace('&', '&amp;').replace('<', '&lt;').replace('>', '&gt;')


def write_elan(recipe, outf, date=None):
    """ELAN document of a recipe: two time slots per line (milliseconds, truncated) and one
    alignable annotation per line on a single `Speakers` tier (aku2elan.py:45-99).
    PARITY UNPINNED (the reference writes through lxml, which is not importable here, and no
    golden `.eaf` exists).  Known deviations from what lxml would emit: a media file whose
    type `mimetypes.guess_type` does not know gets MIME_TYPE="" here, where the reference
    hands lxml a None and fails; attribute values are escaped by `_xml_attr` (&, <, >, ")
    rather than by lxml's serialiser (which also writes ' as-is and numeric character
    references for non-ASCII text under its default ASCII encoding)."""
    from mimetypes import guess_type

    def tag(indent, name, attrs, text=None, close=True):
        a = ''.join(' %s="%s"' % (k, _xml_attr(v)) for k, v in attrs)
        pad = '  ' * indent
        if text is not None:
            outf.write('%s<%s%s>%s</%s>\n' % (pad, name, a, _xml_text(text), name))
        elif close:
            outf.write('%s<%s%s/>\n' % (pad, name, a))
        else:
            outf.write('%s<%s%s>\n' % (pad, name, a))

    outf.write('<ANNOTATION_DOCUMENT xmlns:xsi="http://www.w3.org/2001/XMLSchema-instance"'
               ' xsi:noNamespaceSchemaLocation="http://www.mpi.nl/tools/elan/EAFv2.7.xsd"'
               ' AUTHOR="" DATE="%s" FORMAT="2.7" VERSION="2.7">\n' % _xml_attr(date or date_iso()))
    tag(1, 'HEADER', [('MEDIA_FILE', ''), ('TIME_UNITS', 'milliseconds')], close=False)
    tag(2, 'MEDIA_DESCRIPTOR', [('MEDIA_URL', 'file://' + recipe[0][0]),
                                ('MIME_TYPE', guess_type(recipe[0][0])[0] or ''),
                                ('RELATIVE_MEDIA_URL', '')])
    tag(2, 'PROPERTY', [('NAME', 'lastUsedAnnotationId')], text=str(len(recipe)))
    outf.write('  </HEADER>\n')
    tag(1, 'TIME_ORDER', [], close=False)
    ts = 1
    for line in recipe:
        for t in (line[2], line[3]):
            tag(2, 'TIME_SLOT', [('TIME_SLOT_ID', 'ts%d' % ts), ('TIME_VALUE', str(int(t * 1000)))])
            ts += 1
    outf.write('  </TIME_ORDER>\n')
    tag(1, 'TIER', [('DEFAULT_LOCALE', 'en'), ('LINGUISTIC_TYPE_REF', 'default-lt'), ('TIER_ID', 'Speakers')],
        close=False)
    for n, line in enumerate(recipe, 1):
        tag(2, 'ANNOTATION', [], close=False)
        ref = [('ANNOTATION_ID', 'a%d' % n), ('TIME_SLOT_REF1', 'ts%d' % (2 * n - 1)),
               ('TIME_SLOT_REF2', 'ts%d' % (2 * n))]
        if line[4]:
            tag(3, 'ALIGNABLE_ANNOTATION', ref, close=False)
            tag(4, 'ANNOTATION_VALUE', [], text=line[4])
            outf.write('      </ALIGNABLE_ANNOTATION>\n')
        else:
            tag(3, 'ALIGNABLE_ANNOTATION', ref)
        outf.write('    </ANNOTATION>\n')
    outf.write('  </TIER>\n')
    tag(1, 'LINGUISTIC_TYPE', [('GRAPHIC_REFERENCES', 'false'), ('LINGUISTIC_TYPE_ID', 'default-lt'),
                               ('TIME_ALIGNABLE', 'true')])
    tag(1, 'LOCALE', [('COUNTRY_CODE', 'US'), ('LANGUAGE_CODE', 'en')])
    outf.write('</ANNOTATION_DOCUMENT>\n')


def main_aku2elan(argv=None, stdout=None, date=None):
    out = stdout or sys.stdout
    say = _say_to(out)
    p = argparse.ArgumentParser(description='Converts an AKU recipe to Elan format.')
    p.add_argument('recfile', type=str, help='Specifies the input recipe file')
    p.add_argument('-o', dest='outfile', type=str, default=None, help='Specifies an output file, default stdout.')
    args = p.parse_args(argv)
    say('Reading recipe from:', args.recfile)
    with open(args.recfile, 'r') as f:
        recipe = parse_recipe_ann(f, say)           # same five fields, same regexes (aku2elan.py:20-42)
    if args.outfile is not None:
        say('Writing output to:', args.outfile)
        with open(args.outfile, 'w') as outf:
            write_elan(recipe, outf, date)
    else:
        say('Writing output to: stdout')
        write_elan(recipe, out, date)
    return None


# --------------------------------------------------------------------------- DER
def parse_recipe_der(lines, say):
    """[(audio, [(start, end, speaker)])], entry count.  As in the reference, a change of
    audio file closes the previous group under the NEW file's name (clus-performance.py:30-33);
    the scorer only ever looks at the first group."""
    groups, cur = [], []
    this_file = ''
    total = 0
    for line in lines:
        a, s, e = _AUDIO.search(line), _START.search(line), _END.search(line)
        if not (a and s and e):
            say('Recipe line without recognizable data:')
            say(line)
            continue
        sp = _SPEAKER.search(line)
        audio = a.group(1)
        if audio != this_file:
            if this_file != '':
                groups.append((audio, cur))
                cur = []
            this_file = audio
        cur.append((float(s.group(1)), float(e.group(1)), sp.group(1) if sp else ''))
        total += 1
    groups.append((this_file, cur))
    return groups, total


def labeled_frames(entries, resolution):
    """One label id per `resolution` seconds: -1 = silence, k = index into the returned
    name list.  int((end - start) / resolution) labels per entry, gaps filled with silence,
    in the reference's float operation order (clus-performance.py:54-75)."""
    names, ids, counts = [], [], []
    index = {}
    current = 0.0
    for start, end, label in entries:
        if start > current:
            ids.append(-1)
            counts.append(int((start - current) / resolution))
        if label not in index:
            index[label] = len(names)
            names.append(label)
        ids.append(index[label])
        counts.append(int((end - start) / resolution))
        current = end
    counts = np.maximum(np.array(counts, dtype=np.int64), 0) if counts else np.zeros(0, dtype=np.int64)
    return np.repeat(np.array(ids, dtype=np.int64), counts), names


def der(baseline_entries, proposed_entries, resolution=0.001):
    """(correct, incorrect) frame counts.  Every baseline label (silence included) is
    mapped to the proposed label it co-occurs with most; candidates are ranked by count,
    ties in the order the pairs first appear in time (what the reference's dict-of-dicts
    plus stable sort gives under insertion-ordered dicts)."""
    b, _ = labeled_frames(baseline_entries, resolution)
    p, _ = labeled_frames(proposed_entries, resolution)
    n = min(len(b), len(p))
    b, p = b[:n], p[:n]
    if n == 0:
        return 0, 0
    width = int(p.max()) + 2
    key = (b + 1) * width + (p + 1)
    uniq, first, cnt = np.unique(key, return_index=True, return_counts=True)
    ub = uniq // width
    b_first = {}
    for k, f in zip(ub.tolist(), first.tolist()):
        b_first[k] = min(b_first.get(k, f), f)
    order = sorted(range(len(uniq)), key=lambda i: (b_first[int(ub[i])], int(first[i])))
    order.sort(key=lambda i: int(cnt[i]), reverse=True)          # stable, like sorted(..., reverse=True)
    best = {}
    correct = 0
    for i in order:
        k = int(ub[i])
        if k not in best:
            best[k] = int(uniq[i] % width)
            correct += int(cnt[i])
    return correct, n - correct


def main_clus_performance(argv=None, stdout=None):
    out = stdout or sys.stdout
    say = _say_to(out)
    p = argparse.ArgumentParser(description='Rate a recipe against another, typically to benchmark '
                                'diarization performance.')
    p.add_argument('baseline', type=str, help='Especifies the baseline recipe file.')
    p.add_argument('proposed', type=str, help='Especifies the proposed recipe file, to benchmark.')
    p.add_argument('-o', dest='outfile', type=str, default=None, help='Especifies an output file, default stdout.')
    p.add_argument('-t', dest='threshold', type=float, default=0.25,
                   help='Especifies threshold to determine when a time is incorrect, default 0.25 seconds.')
    for flag in ('-sc', '-si', '-sd', '-ss'):
        p.add_argument(flag, action='store_true', help='accepted and ignored, as in the reference')
    args = p.parse_args(argv)
    say('Reading baseline recipe from:', args.baseline)
    with open(args.baseline, 'r') as f:
        base, _ = parse_recipe_der(f, say)
    say('Reading proposed recipe from:', args.proposed)
    with open(args.proposed, 'r') as f:
        prop, _ = parse_recipe_der(f, say)
    if args.outfile is not None:
        say('Writing output to:', args.outfile)
        open(args.outfile, 'w').close()               # the reference opens it and writes nothing
    else:
        say('Writing output to: stdout')
    say('Threshold:', args.threshold)
    resolution = 0.001
    correct, incorrect = der(base[0][1], prop[0][1], resolution)
    correct, incorrect = float(correct), float(incorrect)
    say('Correct time:', correct * resolution)
    say('Incorrect time:', incorrect * resolution)
    say('Total time:', (incorrect + correct) * resolution)
    say('DER:', incorrect / (incorrect + correct))
    return None
