"""The examples that used to sit in the docstrings of the host modules (see the files of this package), and their runner."""
import doctest
import importlib

MODULES = {
    'descriptors': 'pyimsegm_amd.descriptors', 'graph_cuts': 'pyimsegm_amd.graph_cuts', 'superpixels': 'pyimsegm_amd.superpixels',
    'classification': 'pyimsegm_amd.classification', 'labeling': 'pyimsegm_amd.labeling', 'pipelines': 'pyimsegm_amd.pipelines',
    'utilities_data_io': 'pyimsegm_amd.utilities.data_io',
}


def run_examples(key, on_device=False):
    """runs the examples of one module in that module's namespace (a fresh copy per function, as doctest.testmod does for
    docstrings); ``on_device``: the examples marked ``# doctest: +SKIP`` (they need the GPU) run as well.
    Returns (failed, attempted)."""
    module = importlib.import_module(MODULES[key])
    examples = importlib.import_module('tests.doctests.' + key).EXAMPLES
    parser = doctest.DocTestParser()
    runner = doctest.DocTestRunner(verbose=False)
    for name, text in examples.items():
        if on_device:
            text = text.replace('# doctest: +SKIP +', '# doctest: +').replace('# doctest: +SKIP', '')
        runner.run(parser.get_doctest(text, dict(module.__dict__), '%s.%s' % (key, name), MODULES[key], 0))
    res = runner.summarize(verbose=False)
    return res.failed, res.attempted
