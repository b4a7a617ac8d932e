from . import logger


def dict_value(dictionary, name, default=None, err=False):
    """pixelssl/utils/tool.py:4-16 contract: fetch a key, optionally fatal when missing."""
    if dictionary is None or name not in dictionary:
        if err:
            logger.log_err('Cannot find key: {0}\n'.format(name) if dictionary is not None
                           else 'The given dictionary is None\n')
        return default
    return dictionary[name]
