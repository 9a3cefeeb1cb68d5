"""Share of the code lines of every product python file that occur verbatim (stripped, 12 characters or more, comments excluded) in
the reference's python sources -- the measure VERDICT r2 / r3 quote for `transcription`.  Build container only (/root/reference)."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_lines(path):
    out = []
    for line in open(path, errors='ignore'):
        text = line.strip()
        if text and not text.startswith('#') and len(text) >= 12:
            out.append(text)
    return out


def main():
    ref = set()
    for path in glob.glob('/root/reference/imsegm/**/*.py', recursive=True) + glob.glob('/root/reference/experiments_segmentation/*.py'):
        ref.update(code_lines(path))
    worst = 0.
    for path in sorted(glob.glob(ROOT + '/pyimsegm_amd/**/*.py', recursive=True) + glob.glob(ROOT + '/imsegm/**/*.py', recursive=True)
                       + glob.glob(ROOT + '/gco/*.py')):
        lines = code_lines(path)
        if not lines:
            continue
        share = 100. * sum(1 for text in lines if text in ref) / len(lines)
        worst = max(worst, share)
        print('%-50s %5d lines %5.1f %% shared' % (os.path.relpath(path, ROOT), len(lines), share))
    return 0 if worst < 15. else 1


if __name__ == '__main__':
    sys.exit(main())
