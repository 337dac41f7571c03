"""Readable kernel names for the profile summaries: this image's demanglers (rocprofv3's, c++filt) leave names with a `__bf16` template
argument (`DF16b`) mangled.  Only the pattern our kernels use is handled: _Z<len><name>I<args>E... with args DF16b | f | Li<n>E | Lb<0|1>E."""
import re


def pretty(name):
    m = re.match(r'_Z(\d+)', name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith('I'):
        return base
    rest, args = rest[1:], []
    while rest and not rest.startswith('E'):
        for pat, fn in ((r'DF16b', lambda g: '__bf16'), (r'DF16_', lambda g: '_Float16'), (r'f', lambda g: 'float'),
                        (r'Li(n?\d+)E', lambda g: g.group(1).replace('n', '-')), (r'Lb([01])E', lambda g: 'true' if g.group(1) == '1' else 'false')):
            g = re.match(pat, rest)
            if g:
                args.append(fn(g))
                rest = rest[g.end():]
                break
        else:
            return name
    return f'{base}<{", ".join(args)}>'
