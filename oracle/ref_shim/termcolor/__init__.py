"""Shim: `termcolor.cprint` stand-in (plain print). Used ONLY by oracle/gen_golden.py in the
development container so that /root/reference/pymht/tracker.py can be imported as an oracle."""


def cprint(text, *args, **kwargs):
    kwargs.pop("attrs", None)
    print(text, **{k: v for k, v in kwargs.items() if k in ("end", "sep", "file")})


def colored(text, *args, **kwargs):
    return text
