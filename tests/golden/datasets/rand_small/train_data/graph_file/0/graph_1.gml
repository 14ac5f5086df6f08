graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  edge [
    source 0
    target 1
    capacity 4348
  ]
  edge [
    source 0
    target 8
    capacity 9677
  ]
  edge [
    source 1
    target 0
    capacity 4348
  ]
  edge [
    source 1
    target 2
    capacity 8895
  ]
  edge [
    source 2
    target 1
    capacity 8895
  ]
  edge [
    source 2
    target 3
    capacity 8482
  ]
  edge [
    source 3
    target 2
    capacity 8482
  ]
  edge [
    source 3
    target 4
    capacity 3191
  ]
  edge [
    source 4
    target 3
    capacity 3191
  ]
  edge [
    source 4
    target 5
    capacity 3810
  ]
  edge [
    source 5
    target 4
    capacity 3810
  ]
  edge [
    source 5
    target 6
    capacity 3020
  ]
  edge [
    source 6
    target 5
    capacity 3020
  ]
  edge [
    source 6
    target 7
    capacity 3949
  ]
  edge [
    source 7
    target 6
    capacity 3949
  ]
  edge [
    source 7
    target 8
    capacity 2428
  ]
  edge [
    source 8
    target 0
    capacity 9677
  ]
  edge [
    source 8
    target 7
    capacity 2428
  ]
]
